"""Reference module path `model.evp` (src/model/evp.py) -> the MI355X-native classes of gaviko_amd.model.evp."""
from gaviko_amd.model.evp import *  # noqa: F401,F403
from gaviko_amd.model import evp as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
