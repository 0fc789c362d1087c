"""Reference module path `model.melo` (src/model/melo.py) -> the MI355X-native classes of gaviko_amd.model.melo."""
from gaviko_amd.model.melo import *  # noqa: F401,F403
from gaviko_amd.model import melo as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
