"""The --method plugin surface of the reference (train.py:111-153, eval.py:39-81, inference.py:38-80):
config['model'] (a plain dict or an OmegaConf node; extra keys are swallowed by **kwargs) -> model instance."""
from __future__ import annotations

METHODS = ("gaviko", "linear", "fft", "bitfit", "adaptformer", "dvpt", "evp", "ssf", "melo", "deep_vpt", "shallow_vpt")


def build_model(model_cfg):
    cfg = dict(model_cfg)
    method = cfg["method"]
    if method == "gaviko":
        from .model.gaviko import Gaviko
        return Gaviko(**cfg)
    if method in ("linear", "fft", "bitfit"):
        from .model.vision_transformer import VisionTransformer
        model = VisionTransformer(**cfg)
        for key, value in model.named_parameters():
            if method == "linear":                       # train.py:117-121
                value.requires_grad = "head" in key
            elif method == "bitfit":                     # train.py:131-137
                value.requires_grad = ("bias" in key) or ("head" in key)
        return model
    if method in ("deep_vpt", "shallow_vpt"):
        from .model.vpt import PromptedVisionTransformer
        cfg["deep_prompt"] = method == "deep_vpt"        # train.py:520-523
        return PromptedVisionTransformer(**cfg)
    if method == "adaptformer":
        from .model.adaptformer import AdaptFormer
        return AdaptFormer(**cfg)
    if method == "melo":
        from .model.melo import MeLO
        from .model.vision_transformer import VisionTransformer
        return MeLO(vit=VisionTransformer(**cfg), **cfg)
    if method == "ssf":
        from .model.ssf import ScalingShiftingFeatures
        return ScalingShiftingFeatures(**cfg)
    if method == "dvpt":
        from .model.dvpt import DynamicVisualPromptTuning
        return DynamicVisualPromptTuning(**cfg)
    if method == "evp":
        from .model.evp import ExplicitVisualPrompting
        return ExplicitVisualPrompting(**cfg)
    raise ValueError(f"unknown method {method!r}; expected one of {METHODS}")
