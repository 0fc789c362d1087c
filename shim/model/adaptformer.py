"""Reference module path `model.adaptformer` (src/model/adaptformer.py) -> the MI355X-native classes of gaviko_amd.model.adaptformer."""
from gaviko_amd.model.adaptformer import *  # noqa: F401,F403
from gaviko_amd.model import adaptformer as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
