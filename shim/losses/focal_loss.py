"""Reference module path `losses.focal_loss` (src/losses/focal_loss.py:15-118) -> the fused loss of gaviko_amd.losses (same constructor,
same double clamp + softmax behaviour; one launch for loss and gradient)."""
from gaviko_amd.losses import CrossEntropyLoss, FocalLoss  # noqa: F401
