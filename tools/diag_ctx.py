"""GPU box: accuracy of the attention forward INSIDE a model step -- recompute softmax(q k^T) v in float64 from the saved bf16 qkv of a
layer and compare with the kernel's saved output (ctx) and lse.  Separates kernel error from everything around it."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_model_gpu import build, GAVIKO_CASES, PEFT_CASES
from gaviko_amd.utils import synth

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4_adaptformer_b16_b8"
cases = {c[0]: ("gaviko", c[1], c[2], c[3]) for c in GAVIKO_CASES}
cases.update({c[0]: (c[1], c[2], c[3], c[4]) for c in PEFT_CASES})
method, backbone, B, extra = cases[name]
dev = torch.device("cuda:0")
m, cfg = build(method, backbone, extra, dev)
x = torch.from_numpy(synth.volumes(0, B)).to(dev); y = torch.from_numpy(synth.labels(0, B)).to(dev)
m.train()
torch.nn.functional.cross_entropy(m(x), y).backward(); torch.cuda.synchronize()
eng = m._engine(); ws = eng._ws
H, T = eng.heads, eng.T
inner = H * 64
c = eng.q_scale
for i in (0, eng.depth // 2, eng.depth - 1):
    Ti = eng.Ts[i]; M = B * Ti
    qkv = ws["qkv"][i][:M].double().view(B, Ti, 3, H, 64)
    q, k, v = (qkv[:, :, j].permute(0, 2, 1, 3) for j in range(3))
    s2 = q @ k.transpose(-1, -2)                      # log2 domain (q pre-scaled)
    s = s2 / 1.4426950408889634
    ref = (s.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(M, inner)
    got = ws["ctx"][i][:M].double()
    lse_ref = torch.logsumexp(s, -1)
    lse = ws["lse"][i].double()
    d = got - ref
    print(f"{name} layer {i:2d}: ctx rel rms {d.pow(2).mean().sqrt().item() / ref.pow(2).mean().sqrt().item():.3e}  max|d| {d.abs().max().item():.3e} (|ref| max {ref.abs().max().item():.3f})  "
          f"lse max err {(lse - lse_ref).abs().max().item():.2e}  score max {s.abs().max().item():.1f}  bf16 rounding alone {(ref.bfloat16().double() - ref).pow(2).mean().sqrt().item() / ref.pow(2).mean().sqrt().item():.3e}")
