"""Diagnostic (GPU box): per-tensor gradient error of the HIP path vs a golden fixture."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_model_gpu import build, GAVIKO_CASES, PEFT_CASES, rel
from conftest import golden
from gaviko_amd.utils import synth

name = sys.argv[1] if len(sys.argv) > 1 else "gaviko_t16_b2"
cases = {c[0]: ("gaviko", c[1], c[2], c[3]) for c in GAVIKO_CASES}
cases.update({c[0]: (c[1], c[2], c[3], c[4]) for c in PEFT_CASES})
method, backbone, B, extra = cases[name]
dev = torch.device("cuda:0")
g = golden(name)
m, cfg = build(method, backbone, extra, dev)
x = torch.from_numpy(synth.volumes(0, B)).to(dev); y = torch.from_numpy(synth.labels(0, B)).to(dev)
logits = m(x); torch.nn.functional.cross_entropy(logits, y).backward(); torch.cuda.synchronize()
named = dict(m.named_parameters())
rows = []
for k in g.files:
    if k.startswith("gradnorm/"):
        n = k[9:]; got = named[n].grad.norm().item(); want = float(g[k])
        rows.append((abs(got - want) / max(want, 1e-12), n, got, want))
rows.sort(reverse=True)
print("logits err", rel(logits.detach().cpu().numpy(), g["logits"]))
print("worst 25 grad-norm rel errors:")
for r in rows[:25]:
    print(f"  {r[0]:.3e}  {r[1]:70s} got {r[2]:.4e} want {r[3]:.4e}")
e = np.array([r[0] for r in rows]); print("median", np.median(e), "p90", np.percentile(e, 90), "n", len(e))
print("full-grad rel errors:")
for k in g.files:
    if k.startswith("grad/"):
        print(f"  {rel(named[k[5:]].grad.cpu().numpy(), g[k]):.3e}  {k[5:]}")
