"""CPU: host-side mirror of the reference interface -- registry, parameter schema, freeze rules, train() semantics,
synthetic recipe determinism.  No kernels are launched."""
import numpy as np
import pytest
import torch

import oracle
from gaviko_amd.registry import METHODS, build_model
from gaviko_amd.utils import synth
from gaviko_amd.utils.load_pretrained import mapping_vit

BASE = dict(image_size=160, image_patch_size=16, frames=120, frame_patch_size=12, num_classes=5, channels=1, pool="cls", dim_head=64,
            dropout=0.0, emb_dropout=0.0, backbone="vit-t16")
GAVIKO = dict(num_prompts=32, prompt_latent_dim=20, local_dim=20, local_k=(6, 6, 6), DHW=(10, 10, 10), attn_drop=0.2, proj_drop=0.2,
              freeze_vit=True, share_factor=1, fp16=False)


def test_mapping_vit_table_and_errors():
    assert mapping_vit("vit-b16") == (12, 12, 768, 3072)
    assert mapping_vit("ViT-L16") == (24, 16, 1024, 4096)
    with pytest.raises(ValueError):
        mapping_vit(None)
    with pytest.raises(ValueError):
        mapping_vit("vit-h14")


def test_registry_names():
    assert set(METHODS) == {"gaviko", "linear", "fft", "bitfit", "adaptformer", "dvpt", "evp", "ssf", "melo", "deep_vpt", "shallow_vpt"}
    assert type(build_model(dict(BASE, method="gaviko", **GAVIKO))).__name__ == "Gaviko"
    for m in ("linear", "fft", "bitfit"):
        assert type(build_model(dict(BASE, method=m))).__name__ == "VisionTransformer"
    with pytest.raises(ValueError):
        build_model(dict(BASE, method="nope"))


@pytest.mark.parametrize("backbone,total,trainable", [("vit-t16", 6388289, 274049), ("vit-b16", 89052353, 894401)])
def test_gaviko_schema_and_freeze_rule(backbone, total, trainable):
    cfg = dict(BASE, method="gaviko", **GAVIKO)
    cfg["backbone"] = backbone
    m = build_model(cfg)
    sd = m.state_dict()
    want = oracle.gaviko_param_shapes(cfg, with_alias=True)
    assert set(sd) == set(want) and all(tuple(sd[k].shape) == tuple(want[k]) for k in sd)
    named = dict(m.named_parameters())
    assert len(named) == 442 and sum(p.numel() for p in named.values()) == total            # SURVEY Appendix A [probe]
    tr = [k for k, p in named.items() if p.requires_grad]
    assert len(tr) == 304 and sum(named[k].numel() for k in tr) == trainable
    assert all(oracle.gaviko_trainable(k) == named[k].requires_grad for k in named)
    # aliases share storage with the canonical query projections (gaviko.py:144-145)
    assert sd["transformer.prompt_projs.0.global_query.weight"].data_ptr() == sd["transformer.prompt_projs.0.global_attention.query_proj.weight"].data_ptr()


def test_gaviko_train_override_semantics():
    m = build_model(dict(BASE, method="gaviko", **GAVIKO))
    assert m.train() is None                                   # gaviko.py:513-528 returns None
    assert not m.transformer.training and not m.conv_proj.training and not m.dropout.training
    assert m.transformer.local_attns.training and m.transformer.prompt_projs.training and m.mlp_head.training
    # frozen: the MWSA dropouts are live, the backbone's own nn.Dropout modules sit in eval whatever the config says
    assert m._drop_config() == {"attn_drop": 0.2, "proj_drop": 0.2, "dropout": 0.0, "emb_dropout": 0.0}
    assert m.eval() is None
    assert not m.transformer.local_attns.training
    assert m._drop_config() == {"attn_drop": 0.0, "proj_drop": 0.0, "dropout": 0.0, "emb_dropout": 0.0}
    # freeze_vit=False (gaviko.py:428-434 skipped, 513-528: plain nn.Module.train): everything trains, every dropout follows .training
    u = build_model(dict(BASE, method="gaviko", **dict(GAVIKO, freeze_vit=False, dropout=0.1, emb_dropout=0.1)))
    assert all(p.requires_grad for p in u.parameters())
    u.train()
    assert u.transformer.training and u.conv_proj.training and u.dropout.training
    assert u._drop_config() == {"attn_drop": 0.2, "proj_drop": 0.2, "dropout": 0.1, "emb_dropout": 0.1}
    u.eval()
    assert u._drop_config() == {"attn_drop": 0.0, "proj_drop": 0.0, "dropout": 0.0, "emb_dropout": 0.0}


def test_linear_and_bitfit_freeze_rules():
    m = build_model(dict(BASE, method="linear"))
    tr = [k for k, p in m.named_parameters() if p.requires_grad]
    assert tr == ["mlp_head.weight", "mlp_head.bias"]
    m = build_model(dict(BASE, method="bitfit"))
    named = dict(m.named_parameters())
    assert all(p.requires_grad == (("bias" in k) or ("head" in k)) for k, p in named.items())
    assert set(m.state_dict()) == set(oracle.vit_param_shapes(dict(BASE, method="linear")))


@pytest.mark.parametrize("lora_layer", [None, [0, 5, 11]])
def test_melo_schema_and_lora_layer_subset(lora_layer):
    """melo.py:53-68: `lora_layer` wraps only the listed layers' to_qkv (keys `...to_qkv.{qkv,linear_a_q,...}.weight`); every other layer keeps
    its plain `...to_qkv.weight`.  The parameter schema equals the oracle's (which is checked against the reference's own state_dict)."""
    cfg = dict(BASE, method="melo", r=4, alpha=8, lora_layer=lora_layer)
    m = build_model(cfg)
    named = dict(m.named_parameters())
    want = oracle.SHAPES["melo"](cfg)
    assert set(named) == set(want) and all(tuple(named[k].shape) == tuple(want[k]) for k in named)
    wrapped = lora_layer or list(range(12))
    for i in range(12):
        assert (f"lora_vit.transformer.layers.{i}.0.to_qkv.qkv.weight" in named) == (i in wrapped)
        assert (f"lora_vit.transformer.layers.{i}.0.to_qkv.weight" in named) == (i not in wrapped)
    tr = {k for k, p in named.items() if p.requires_grad}
    assert tr == {k for k in named if oracle.trainable("melo", k)} and len(tr) == 4 * len(wrapped) + 2


def test_mwsa_mask_property_equals_oracle():
    m = build_model(dict(BASE, method="gaviko", **GAVIKO))
    mask = m.transformer.local_attns[0].mask
    assert mask.shape == (1, 1000, 1000)
    assert torch.equal(mask[0], oracle.window_mask((10, 10, 10), (6, 6, 6)))


def test_containers_refuse_to_run_and_cpu_input_fails_loudly():
    from gaviko_amd import lib
    m = build_model(dict(BASE, method="gaviko", **GAVIKO))
    with pytest.raises(lib.GavikoHipError):
        m.transformer.attns[0](torch.zeros(1, 4, 192))
    with pytest.raises(lib.GavikoHipError):
        m(torch.zeros(1, 1, 120, 160, 160))


def test_synth_recipe_is_deterministic_and_bounded():
    a, b = synth.volume(3), synth.volume(3)
    assert a.shape == (1, 120, 160, 160) and np.array_equal(a, b) and a.min() >= 0 and a.max() < 1
    assert not np.array_equal(a, synth.volume(4))
    w = synth.fill_param("transformer.attns.0.to_qkv.weight", (2304, 768))
    assert abs(float(w.std()) * np.sqrt(768) - 1.0) < 0.02
    assert np.array_equal(synth.labels(3, 4), np.array([3, 4, 0, 1]))


def test_ssf_schema_and_freeze_rule():
    """ScalingShiftingFeatures mirror: state_dict keys / shapes of the reference (checked against the fixture's trainable list, which
    came from the reference class itself) and the freeze rule of ssf.py:192-197."""
    from conftest import golden
    cfg = dict(BASE, method="ssf", freeze_vit=True)
    m = build_model(cfg)
    assert type(m).__name__ == "ScalingShiftingFeatures"
    sd = m.state_dict()
    want = oracle.ssf_param_shapes(cfg)
    assert list(sd) == list(want) and all(tuple(sd[k].shape) == tuple(want[k]) for k in sd)      # same registration order too
    named = dict(m.named_parameters())
    tr = sorted(k for k, p in named.items() if p.requires_grad)
    assert tr == sorted(str(k) for k in golden("ssf_t16_b2")["meta/trainable"])
    assert all(oracle.ssf_trainable(k) == named[k].requires_grad for k in named)
    assert m.train() is None and not m.transformer.training and m.mlp_head.training


def test_evp_schema_freeze_rule_and_highpass_operator():
    """ExplicitVisualPrompting mirror (state_dict keys / order / freeze rule of the reference) and the host-built linear operator that
    replaces PromptGenerator.fft: it reproduces the oracle's torch.fft restatement -- including the reference's mask-on-(D,H) indexing
    and its all-axes fftshift -- to fp32 round-off."""
    from conftest import golden
    from gaviko_amd.engine import evp_highpass_operator
    cfg = dict(BASE, method="evp", freeze_vit=True)
    m = build_model(cfg)
    assert type(m).__name__ == "ExplicitVisualPrompting"
    sd = m.state_dict()
    want = oracle.evp_param_shapes(cfg)
    assert list(sd) == list(want) and all(tuple(sd[k].shape) == tuple(want[k]) for k in sd)
    named = dict(m.named_parameters())
    tr = sorted(k for k, p in named.items() if p.requires_grad)
    assert tr == sorted(str(k) for k in golden("evp_t16_b2")["meta/trainable"]) and len(tr) == 32
    assert all(oracle.evp_trainable(k) == named[k].requires_grad for k in named)
    assert m.train() is None and not m.transformer.training and m.prompt_generator.training
    for D, H, W in ((24, 32, 32), (12, 16, 16), (120, 160, 160)):
        x = torch.rand(1, 1, D, H, W, generator=torch.Generator().manual_seed(D))
        hp, dm = evp_highpass_operator(D, H, W, 0.25)
        got = torch.where(torch.from_numpy(dm).bool()[None, None, :, None, None], torch.einsum("ik,bcdkj->bcdij", torch.from_numpy(hp), x).abs(), x.abs())
        assert (got - oracle.fft_highpass(x, 0.25)).abs().max() < 2e-6
    assert int(evp_highpass_operator(120, 160, 160, 0.25)[1].sum()) == 80       # 80 of the 120 depth slices are filtered (quirk 17)


def test_dvpt_schema_and_freeze_rule():
    from conftest import golden
    cfg = dict(BASE, method="dvpt", num_prompts=50, freeze_vit=True)
    m = build_model(cfg)
    assert type(m).__name__ == "DynamicVisualPromptTuning"
    sd = m.state_dict()
    want = oracle.dvpt_param_shapes(cfg)
    assert list(sd) == list(want) and all(tuple(sd[k].shape) == tuple(want[k]) for k in sd)
    named = dict(m.named_parameters())
    tr = sorted(k for k, p in named.items() if p.requires_grad)
    assert tr == sorted(str(k) for k in golden("dvpt_t16_b2")["meta/trainable"]) and len(tr) == 64
    assert all(oracle.dvpt_trainable(k) == named[k].requires_grad for k in named)
    assert m.train() is None and not m.transformer.layers[0][0].attn.training and m.transformer.layers[0][0].prompt_proj.training


def test_product_source_hash_ignores_diag_blocks_and_comments():
    """bench.py attaches profiles/r0N_pmc_traffic.json only when the GEMM sources' hash matches; the hash covers what the PRODUCT build
    compiles (round 3: a diag-only edit withheld roofline.traffic from the driver's line)."""
    from gaviko_amd.utils.srchash import gemm_source_hash, product_text
    base = "int f(int x) {\n  return x + 1;   // add one\n}\n"
    diag = "int f(int x) {\n#ifdef GVK_DIAG\n  if (getenv(\"A\")) return 0;\n#if FOO\n  x = 2;\n#endif\n#endif\n  /* block\n comment */ return x + 1;\n}\n"
    assert product_text(base) == product_text(diag)
    both = "#ifdef GVK_DIAG\nint a;\n#else\nint b;\n#endif\n#ifndef GVK_DIAG\nint c;\n#else\nint d;\n#endif\n#if FOO\nint e;\n#else\nint g;\n#endif\n"
    assert product_text(both).split("\n") == ["int b;", "int c;", "#if FOO", "int e;", "#else", "int g;", "#endif"]
    assert product_text(base) != product_text(base.replace("x + 1", "x + 2"))
    h = gemm_source_hash()
    assert len(h) == 16 and h == gemm_source_hash()


def test_pmc_traffic_clusters_split_shapes_of_one_instantiation():
    """tools/pmc_traffic.py: one GEMM instantiation serving two shapes whose read traffic differs by 1.3x (fc1 dgrad K = 3072 and qkv dgrad
    K = 2304 share the plain-store kernel) must come out as two clusters -- bench.py attaches `roofline.traffic` only then -- while the
    launch-to-launch spread of one shape (a few per cent) must not split it."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("pmc_traffic", os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools", "pmc_traffic.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    a = [24800 + 37 * i for i in range(72)]            # KiB as counted (reads are doubled afterwards): +-5 %
    b = [33000 + 29 * i for i in range(72)]
    reads = [v for pair in zip(a, b) for v in pair]    # the two shapes alternate in launch order
    writes = [12400.0] * len(reads)
    cl = mod.clusters(reads, writes)
    assert [c["launches"] for c in cl] == [72, 72]
    assert cl[0]["read_bytes"] < cl[1]["read_bytes"] and abs(cl[1]["read_bytes"] / cl[0]["read_bytes"] - 1.33) < 0.05
    one = mod.clusters(a, [12400.0] * len(a))
    assert len(one) == 1 and one[0]["launches"] == 72


def test_pmc_traffic_is_attached_only_to_the_shape_it_was_measured_on(tmp_path):
    """Round 4 shipped a cfg5 line carrying the cfg2 kernel's PMC bytes: bench.py paired traffic clusters with this run's shapes by rank
    order.  Now tools/pmc_traffic.py stores [M, N, K] with every cluster and bench.pmc_traffic attaches a figure only on an exact match:
    a PMC file taken at cfg2 must yield `traffic` for cfg2's fc1 dgrad and NOTHING but a note for cfg5's."""
    import importlib.util, json, os, sys
    root = os.path.dirname(os.path.dirname(__file__))
    sys.path.insert(0, root)
    import bench
    spec = importlib.util.spec_from_file_location("pmc_traffic", os.path.join(root, "tools", "pmc_traffic.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    from gaviko_amd.engine_common import _EPI_NAMES
    kname = "gemm_nt_kernel<128, 128, 5, 64, false, 3, 4>(GemmArgs)"
    assert bench.gemm_kernel_epilogue(kname) == 5 and bench.gemm_kernel_epilogue("gemm8p_kernel<2, 0>(GemmArgs)") == 2
    assert bench.gemm_kernel_epilogue("param_grads_kernel<2>(PgArgs)") is None
    # the instantiation as the current build prints it (trailing split-K flag), and the panel tile
    assert bench.gemm_kernel_epilogue("gemm_nt_kernel<128, 128, 6, 64, false, 3, 4, false>(GemmArgs)") == 6
    assert bench.gemm_kernel_epilogue("gemm_nt_kernel<64, 128, 4, 64, false, 4, 4, true>(GemmArgs)") == 4
    assert mod.kernel_family("gemm_nt_kernel<64, 128, 4, 64, false, 4, 4, false>(GemmArgs)") == "panels"
    kernels = {kname: {"launches": 144, "clusters": [{"launches": 72, "total_bytes": 63526960}, {"launches": 72, "total_bytes": 80435275}]},
               "param_grads_kernel<2>(PgArgs)": {"launches": 144, "clusters": [{"launches": 144, "total_bytes": 49017128}]}}
    cfg2 = {"workload": {"backbone": "vit-b16", "method": "gaviko", "batch": 4, "precision": "bf16"},
            "classes": {"gemm_nt_bf16[store_f32] M=4132 N=768 K=3072": [4132, 768, 3072], "gemm_nt_bf16[store_f32] M=4132 N=768 K=2304": [4132, 768, 2304],
                        "gemm_nt_bf16[store_bf16] M=4132 N=768 K=768": [4132, 768, 768]}}
    mod.attach_shapes(kernels, cfg2, bench.gemm_kernel_epilogue, bench.gemm_alg_bytes, _EPI_NAMES)
    cl = kernels[kname]["clusters"]
    assert cl[0]["shape"] == [4132, 768, 2304] and cl[1]["shape"] == [4132, 768, 3072]          # ascending bytes <-> ascending K
    assert "shape" not in kernels["param_grads_kernel<2>(PgArgs)"]["clusters"][0]
    doc = {"gemm_source_sha": bench.gemm_source_hash(), "workload": cfg2["workload"], "kernels": kernels}
    path = tmp_path / "pmc.json"
    path.write_text(json.dumps(doc))
    name2 = "gemm_nt_bf16[store_f32] M=4132 N=768 K=3072"
    r = bench.pmc_traffic(name2, {name2: {"shape": [4132, 768, 3072]}, "gemm_nt_bf16[store_f32] M=4132 N=768 K=2304": {"shape": [4132, 768, 2304]}}, path=str(path))
    assert r["traffic"] == 80435275 and r["algorithmic_bytes"] == bench.gemm_alg_bytes(5, [4132, 768, 3072])
    # cfg5: same instantiation, two classes as well (so rank-order pairing would have "matched"), other M / N / K
    name5 = "gemm_nt_bf16[store_f32] M=2066 N=1024 K=4096"
    r5 = bench.pmc_traffic(name5, {name5: {"shape": [2066, 1024, 4096]}, "gemm_nt_bf16[store_f32] M=2066 N=1024 K=3072": {"shape": [2066, 1024, 3072]}}, path=str(path))
    assert "traffic" not in r5 and "withheld" in r5["traffic_note"]
    # a file without recorded shapes (rounds 1-4) gives no number either
    for c in cl:
        c.pop("shape")
    path.write_text(json.dumps(doc))
    r_old = bench.pmc_traffic(name2, {name2: {"shape": [4132, 768, 3072]}}, path=str(path))
    assert "traffic" not in r_old and "no [M, N, K]" in r_old["traffic_note"]
    # clusters that moved fewer bytes than the paired shape needs are left unlabelled (the pairing cannot be right)
    tiny = {kname: {"clusters": [{"total_bytes": 1000}, {"total_bytes": 2000}]}}
    mod.attach_shapes(tiny, cfg2, bench.gemm_kernel_epilogue, bench.gemm_alg_bytes, _EPI_NAMES)
    assert all("shape" not in c for c in tiny[kname]["clusters"])
