"""gaviko_amd -- MI355X (gfx950) native kernels for GAViKO's 3D-ViT training hot path.

Layout: csrc/ (HIP kernels + the C-ABI of include/gaviko_hip.h), lib.py/ops.py (ctypes launchers),
model/ (host-side mirror of the reference's `model.*` nn.Module / --method surface).
"""
__version__ = "0.1.0"
