"""CPU: with shim/ first on sys.path the reference's own import statements (src/train.py:5-14,21; src/eval.py:3-12,19) resolve to the
gaviko_amd classes, `utils.logging` still resolves to a reference-side file, and all 11 --method branches of train.py:111-153 build.
Runs in a fresh interpreter: `model`, `utils`, `data`, `losses` are too generic to leave in this process's module cache."""
import os
import subprocess
import sys
import textwrap

from conftest import ROOT

SCRIPT = textwrap.dedent('''
    import sys
    # the statements of src/train.py:5-14 and :21 (third-party imports of the file left out), typed here
    from data.dataset import CustomDataset
    from model.gaviko import Gaviko
    from model.adaptformer import AdaptFormer
    from model.vision_transformer import VisionTransformer
    from model.dvpt import DynamicVisualPromptTuning
    from model.evp import ExplicitVisualPrompting
    from model.ssf import ScalingShiftingFeatures
    from model.melo import MeLO
    from model.vpt import PromptedVisionTransformer
    from utils.logging import MemoryUsageLogger, analyze_model_computation
    from losses.focal_loss import FocalLoss
    # src/eval.py:3,19
    from data.dataset import CustomDataset, CustomDatasetPrediction
    from utils.load_pretrained import load_vanilla_pretrain_with_adapters

    import gaviko_amd.model.gaviko, gaviko_amd.model.vision_transformer, gaviko_amd.model.vpt, gaviko_amd.model.adaptformer
    import gaviko_amd.model.melo, gaviko_amd.model.ssf, gaviko_amd.model.dvpt, gaviko_amd.model.evp, gaviko_amd.losses, gaviko_amd.data
    assert Gaviko is gaviko_amd.model.gaviko.Gaviko
    assert VisionTransformer is gaviko_amd.model.vision_transformer.VisionTransformer
    assert PromptedVisionTransformer is gaviko_amd.model.vpt.PromptedVisionTransformer
    assert AdaptFormer is gaviko_amd.model.adaptformer.AdaptFormer and MeLO is gaviko_amd.model.melo.MeLO
    assert ScalingShiftingFeatures is gaviko_amd.model.ssf.ScalingShiftingFeatures
    assert DynamicVisualPromptTuning is gaviko_amd.model.dvpt.DynamicVisualPromptTuning
    assert ExplicitVisualPrompting is gaviko_amd.model.evp.ExplicitVisualPrompting
    assert FocalLoss is gaviko_amd.losses.FocalLoss and CustomDataset is gaviko_amd.data.CustomDataset
    assert MemoryUsageLogger == "reference-side logging module"          # utils.logging came from the OTHER utils/ directory

    # the factory of src/train.py:111-153, branch for branch
    base = dict(image_size=160, image_patch_size=16, frames=120, frame_patch_size=12, num_classes=5, channels=1, pool="cls", dim_head=64,
                dropout=0.1, emb_dropout=0.1, backbone="vit-t16", fp16=False, depth=12, heads=3, dim=192, mlp_dim=768,
                num_prompts=8, prompt_dim=64, prompt_dropout=0.1, deep_prompt=True, r=4, alpha=4, lora_layer=None,
                prompt_latent_dim=20, local_dim=20, local_k=(6, 6, 6), DHW=(10, 10, 10), attn_drop=0.2, proj_drop=0.2, freeze_vit=True,
                share_factor=1)
    built = {}
    for method in ("gaviko", "linear", "fft", "adaptformer", "bitfit", "dvpt", "evp", "ssf", "melo", "deep_vpt", "shallow_vpt"):
        config = {"model": dict(base, method=method)}
        if method == "gaviko":
            model = Gaviko(**config["model"])
        elif method == "linear":
            model = VisionTransformer(**config["model"])
            for key, value in model.named_parameters():
                value.requires_grad = "head" in key
        elif method == "fft":
            model = VisionTransformer(**config["model"])
        elif method == "adaptformer":
            model = AdaptFormer(**config["model"])
        elif method == "bitfit":
            model = VisionTransformer(**config["model"])
            for key, value in model.named_parameters():
                value.requires_grad = ("bias" in key) or ("head" in key)
        elif method == "dvpt":
            model = DynamicVisualPromptTuning(**config["model"])
        elif method == "evp":
            model = ExplicitVisualPrompting(**config["model"])
        elif method == "ssf":
            model = ScalingShiftingFeatures(**config["model"])
        elif method == "melo":
            model = MeLO(vit=VisionTransformer(**config["model"]), **config["model"])
        else:
            config["model"]["deep_prompt"] = method == "deep_vpt"
            model = PromptedVisionTransformer(**config["model"])
        built[method] = sum(p.numel() for p in model.parameters() if p.requires_grad)
    assert len(built) == 11 and all(v > 0 for v in built.values()), built
    assert built["linear"] == 192 * 5 + 5
    print("SHIM-OK", built)
''')


def test_reference_imports_resolve_through_the_shim(tmp_path):
    ref_like = tmp_path / "src"                      # stands in for the reference's src/: only its utils/logging.py matters here
    (ref_like / "utils").mkdir(parents=True)
    (ref_like / "utils" / "logging.py").write_text("MemoryUsageLogger = analyze_model_computation = setup_logging = 'reference-side logging module'\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "shim"), ROOT, str(ref_like)]), PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", SCRIPT], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SHIM-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_shim_wins_when_the_reference_src_is_first_on_sys_path(tmp_path):
    """`python src/train.py` puts src/ at sys.path[0], AHEAD of everything on PYTHONPATH.  The shim still wins because the reference's
    model/ data/ losses/ utils/ directories carry no __init__.py: they are namespace-package candidates, and a regular package found
    LATER on the path takes precedence over a namespace portion found earlier (import system: a path entry with <name>/__init__.py ends
    the search, one without only records a portion).  This stand-in mirrors that layout with decoy modules that must never be imported."""
    ref_like = tmp_path / "src"
    for pkg, mods in (("model", ("gaviko", "vision_transformer", "vpt", "adaptformer", "melo", "ssf", "dvpt", "evp")),
                      ("data", ("dataset",)), ("losses", ("focal_loss",)), ("utils", ("load_pretrained",))):
        (ref_like / pkg).mkdir(parents=True)                     # no __init__.py, like /root/reference/src/*
        for m in mods:
            (ref_like / pkg / f"{m}.py").write_text("raise ImportError('the reference-side module was imported instead of the shim')\n")
    (ref_like / "utils" / "logging.py").write_text("MemoryUsageLogger = analyze_model_computation = setup_logging = 'reference-side logging module'\n")
    (ref_like / "train_like.py").write_text("import sys, os\nassert os.path.abspath(sys.path[0]) == os.path.dirname(os.path.abspath(__file__)), sys.path[:3]\n" + SCRIPT)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "shim"), ROOT]), PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, str(ref_like / "train_like.py")], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SHIM-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
