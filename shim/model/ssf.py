"""Reference module path `model.ssf` (src/model/ssf.py) -> the MI355X-native classes of gaviko_amd.model.ssf."""
from gaviko_amd.model.ssf import *  # noqa: F401,F403
from gaviko_amd.model import ssf as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
