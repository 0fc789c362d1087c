"""Pretrained-weight / checkpoint interchange (gaviko_amd/utils/load_pretrained.py) against fixtures produced by the reference's own
load_pretrain (tools/gen_golden.py pretrain) -- host logic, runs without a GPU."""
import os

import numpy as np
import pytest
import torch

from gaviko_amd.utils import load_pretrained as lp

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CFG = dict(image_size=160, image_patch_size=16, frames=120, frame_patch_size=12, num_classes=5, channels=1, backbone="vit-t16")


def _fixture(tag):
    g = np.load(os.path.join(GOLD, f"pretrain_convert{tag}.npz"))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("in/")}
    want = {k[4:]: g[k] for k in g.files if k.startswith("out/")}
    return g, sd, want


@pytest.mark.parametrize("tag", ["", "_n216_d4"])
def test_converter_matches_reference_function(tag):
    g, sd, want = _fixture(tag)
    out = lp.convert_timm_state_dict(sd, int(g["meta/num_patches"]), int(g["meta/depth_dim"]))
    assert list(out.keys()) == list(want.keys())                     # same keys, same insertion order
    for k, v in out.items():
        assert v.shape == want[k].shape, k
        assert np.array_equal(v.numpy(), want[k]), k                  # same torch calls on the same data: bit-exact
    # the quirks the reference has, kept: qkv.bias carried to a destination that does not exist, head / pre_logits dropped
    assert "transformer.attns.0.to_qkv.bias" in out and not any(k.startswith(("head", "pre_logits")) for k in out)
    n = round(int(g["meta/num_patches"]) ** (1 / 3))
    assert out["pos_embedding"].shape == (1, 1 + n ** 3, 16)
    assert out["conv_proj.0.weight"].shape == (16, 1, int(g["meta/depth_dim"]), 16, 16)


def test_load_pretrain_reads_the_file_the_reference_leaves_behind(tmp_path):
    g, sd, want = _fixture("")
    name = str(g["meta/saved_as"][0])
    assert lp.TIMM_NAMES["vit-t16"] == name
    with pytest.raises(FileNotFoundError, match="no network"):
        lp.load_pretrain("vit_t16", 1000, 12, str(tmp_path))
    torch.save(sd, os.path.join(tmp_path, name))
    out = lp.load_pretrain("vit_t16", 1000, 12, str(tmp_path))       # '_' spelling is normalised like load_pretrained.py:10
    assert all(np.array_equal(out[k].numpy(), want[k]) for k in want)
    with pytest.raises(ValueError):
        lp.load_pretrain("resnet50", 1000, 12, str(tmp_path))
    cfg = {"model": dict(CFG)}
    out2 = lp.load_vanilla_pretrain("vit-t16", cfg, save_dir=str(tmp_path))
    assert np.array_equal(out2["pos_embedding"].numpy(), want["pos_embedding"])
    ck = os.path.join(tmp_path, "adapters.pt")
    torch.save({"mlp_head.weight": torch.ones(5, 16), "pos_embedding": torch.zeros(1, 1001, 16)}, ck)
    merged = lp.load_vanilla_pretrain_with_adapters("vit-t16", cfg, ck, save_dir=str(tmp_path))
    assert merged["pos_embedding"].abs().sum() == 0 and "mlp_head.weight" in merged and "cls_token" in merged   # checkpoint wins


def _timm_like(dim, depth, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    sd = {"cls_token": r(1, 1, dim), "pos_embed": r(1, 197, dim), "patch_embed.proj.weight": r(dim, 3, 16, 16), "patch_embed.proj.bias": r(dim),
          "norm.weight": r(dim), "norm.bias": r(dim), "head.weight": r(9, dim), "head.bias": r(9)}
    for i in range(depth):
        b = f"blocks.{i}."
        sd.update({b + "norm1.weight": r(dim), b + "norm1.bias": r(dim), b + "attn.qkv.weight": r(3 * dim, dim), b + "attn.qkv.bias": r(3 * dim),
                   b + "attn.proj.weight": r(dim, dim), b + "attn.proj.bias": r(dim), b + "norm2.weight": r(dim), b + "norm2.bias": r(dim),
                   b + "mlp.fc1.weight": r(4 * dim, dim), b + "mlp.fc1.bias": r(4 * dim), b + "mlp.fc2.weight": r(dim, 4 * dim),
                   b + "mlp.fc2.bias": r(dim)})
    return sd


def test_constructors_load_the_backbone_like_the_reference(tmp_path, monkeypatch):
    """Gaviko names its blocks attns/mlps and receives every backbone tensor; the plain ViT names them layers.i.j, so under strict=False
    only the embedding / final norm arrive -- unless the caller opts into remap_blocks_to_layers."""
    from gaviko_amd.model.gaviko import Gaviko
    from gaviko_amd.model.vision_transformer import VisionTransformer
    sd = _timm_like(192, 12)
    torch.save(sd, os.path.join(tmp_path, lp.TIMM_NAMES["vit-t16"]))
    monkeypatch.setenv("GAVIKO_PRETRAINED_DIR", str(tmp_path))
    conv = lp.convert_timm_state_dict(sd, 1000, 12)
    gv = Gaviko(**CFG, num_prompts=8)
    st = gv.state_dict()
    for k in ("transformer.attns.3.to_qkv.weight", "transformer.mlps.11.net.4.bias", "conv_proj.0.weight", "pos_embedding", "cls_token",
              "transformer.norm.weight"):
        assert torch.equal(st[k], conv[k]), k
    vt = VisionTransformer(**CFG)
    st = vt.state_dict()
    assert torch.equal(st["pos_embedding"], conv["pos_embedding"]) and torch.equal(st["conv_proj.0.weight"], conv["conv_proj.0.weight"])
    assert not torch.equal(st["transformer.layers.3.0.to_qkv.weight"], conv["transformer.attns.3.to_qkv.weight"])       # dropped, as in the reference
    missing, unexpected = vt.load_state_dict(lp.remap_blocks_to_layers(conv), strict=False)
    assert sorted(unexpected) == [f"transformer.layers.{i}.0.to_qkv.bias" for i in sorted(range(12), key=str)] or \
        set(unexpected) == {f"transformer.layers.{i}.0.to_qkv.bias" for i in range(12)}
    assert set(missing) == {"mlp_head.weight", "mlp_head.bias"}
    assert torch.equal(vt.state_dict()["transformer.layers.3.0.to_qkv.weight"], conv["transformer.attns.3.to_qkv.weight"])
    monkeypatch.delenv("GAVIKO_PRETRAINED_DIR")
    monkeypatch.chdir(tmp_path / "..")                                   # no ./pretrained here: construction falls back to random init
    VisionTransformer(**CFG)


def test_trainable_only_checkpoint_roundtrip(tmp_path):
    from gaviko_amd.model.gaviko import Gaviko
    torch.manual_seed(0)
    a = Gaviko(**CFG, num_prompts=8, freeze_vit=True)
    names = lp.tuning_param_names(a)
    assert names and all(not n.startswith(("transformer.attns", "transformer.mlps", "conv_proj", "pos_embedding")) for n in names)
    path = lp.save_trainable(a, str(tmp_path), "gaviko", "vit-t16", 7, 0.91234)
    assert path.endswith(os.path.join("experiments", "gaviko", "gaviko_vit_t16_best_model_epoch7_acc0.9123.pt"))   # train.py:466-469
    ck = torch.load(path, map_location="cpu")
    assert sorted(ck.keys()) == sorted(names)                          # train.py:479-483: exactly the requires_grad names
    torch.manual_seed(1)
    b = Gaviko(**CFG, num_prompts=8, freeze_vit=True)
    missing, unexpected = b.load_state_dict(ck, strict=False)           # eval.py:92
    assert not unexpected and all(m not in names for m in missing)
    sa, sb = a.state_dict(), b.state_dict()
    assert all(torch.equal(sa[k], sb[k]) for k in names)
