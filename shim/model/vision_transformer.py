"""Reference module path `model.vision_transformer` (src/model/vision_transformer.py) -> the MI355X-native classes of gaviko_amd.model.vision_transformer."""
from gaviko_amd.model.vision_transformer import *  # noqa: F401,F403
from gaviko_amd.model import vision_transformer as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
