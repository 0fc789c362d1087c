"""Reference module path `model.gaviko` (src/model/gaviko.py) -> the MI355X-native classes of gaviko_amd.model.gaviko."""
from gaviko_amd.model.gaviko import *  # noqa: F401,F403
from gaviko_amd.model import gaviko as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
