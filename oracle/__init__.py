"""ORACLE -- test infrastructure only.

CPU restatement of the reference's hot path, used ONLY as a checker by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg.  Nothing under gaviko_amd/ imports it.
"""
from .vit_ref import mapping_vit, vit_forward, vit_param_shapes  # noqa: F401
from .gaviko_ref import gaviko_forward, gaviko_param_shapes, gaviko_trainable, window_mask  # noqa: F401
from .peft_ref import (adaptformer_forward, adaptformer_param_shapes, adaptformer_trainable, melo_forward,  # noqa: F401
                       melo_param_shapes, melo_trainable, vpt_forward, vpt_param_shapes, vpt_trainable)
from .ssf_ref import ssf_forward, ssf_param_shapes, ssf_trainable  # noqa: F401
from .dvpt_ref import dvpt_forward, dvpt_param_shapes, dvpt_trainable  # noqa: F401
from .evp_ref import evp_forward, evp_param_shapes, evp_trainable, fft_highpass  # noqa: F401
from .losses_ref import cross_entropy, focal_loss  # noqa: F401
from . import data_ref  # noqa: F401

FORWARD = {
    "gaviko": gaviko_forward, "linear": vit_forward, "fft": vit_forward, "bitfit": vit_forward,
    "deep_vpt": vpt_forward, "shallow_vpt": vpt_forward, "adaptformer": adaptformer_forward, "melo": melo_forward,
    "ssf": ssf_forward, "dvpt": dvpt_forward, "evp": evp_forward,
}
SHAPES = {
    "gaviko": gaviko_param_shapes, "linear": vit_param_shapes, "fft": vit_param_shapes, "bitfit": vit_param_shapes,
    "deep_vpt": vpt_param_shapes, "shallow_vpt": vpt_param_shapes, "adaptformer": adaptformer_param_shapes,
    "melo": melo_param_shapes, "ssf": ssf_param_shapes, "dvpt": dvpt_param_shapes, "evp": evp_param_shapes,
}


def trainable(method: str, name: str, cfg: dict = None) -> bool:
    """requires_grad of parameter `name` as the reference's constructors leave it.  `cfg` matters only for freeze_vit=False on the classes
    that freeze by default: the freeze loops (adaptformer.py:163-168 and its copies) are skipped and every parameter keeps nn.Parameter's
    default requires_grad=True."""
    if cfg is not None and cfg.get("freeze_vit") is False and method in ("adaptformer", "gaviko", "dvpt", "evp", "deep_vpt", "shallow_vpt", "ssf"):
        return True
    if method == "gaviko":
        return gaviko_trainable(name)
    if method == "linear":       # train.py:117-121
        return "head" in name
    if method == "bitfit":       # train.py:131-137
        return "bias" in name or "head" in name
    if method == "fft":
        return True
    if method in ("deep_vpt", "shallow_vpt"):
        return vpt_trainable(name)
    if method == "adaptformer":
        return adaptformer_trainable(name)
    if method == "melo":
        return melo_trainable(name)
    if method == "ssf":
        return ssf_trainable(name)
    if method == "dvpt":
        return dvpt_trainable(name)
    if method == "evp":
        return evp_trainable(name)
    raise ValueError(method)
