"""Reference module path `model.dvpt` (src/model/dvpt.py) -> the MI355X-native classes of gaviko_amd.model.dvpt."""
from gaviko_amd.model.dvpt import *  # noqa: F401,F403
from gaviko_amd.model import dvpt as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
