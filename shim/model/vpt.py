"""Reference module path `model.vpt` (src/model/vpt.py) -> the MI355X-native classes of gaviko_amd.model.vpt."""
from gaviko_amd.model.vpt import *  # noqa: F401,F403
from gaviko_amd.model import vpt as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
