"""GPU box: how much does a model-level parity figure move between numerically equivalent builds of the attention forward?
Runs one fixture's forward under the kernel's tile-size / row-sum variants (all of them exact to rounding) and prints the logits error
against the golden fixture for each -- the spread is the realisation noise of that figure."""
import sys, os
os.environ.setdefault("GAVIKO_HIP_DIAG", "1")      # kernel variants / A/B switches live in the measurement build (python -m gaviko_amd.build --diag)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_model_gpu import build, GAVIKO_CASES, PEFT_CASES, rel
from conftest import golden
from gaviko_amd.utils import synth

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4_adaptformer_b16_b8"
cases = {c[0]: ("gaviko", c[1], c[2], c[3]) for c in GAVIKO_CASES}
cases.update({c[0]: (c[1], c[2], c[3], c[4]) for c in PEFT_CASES})
method, backbone, B, extra = cases[name]
dev = torch.device("cuda:0")
g = golden(name)
m, cfg = build(method, backbone, extra, dev)
x = torch.from_numpy(synth.volumes(0, B)).to(dev)
eng = m._engine()
m.eval()
for kb in ("96", "128"):
    for var in ("0", "1"):
        os.environ["GAVIKO_HIP_ATTN_KB"], os.environ["GAVIKO_HIP_ATTN_VAR"] = kb, var
        eng._graphs.clear(); eng._calls.clear()
        with torch.no_grad():
            lg = m(x).detach().cpu().numpy()
        want = g["logits"][:B]
        print(f"{name} KB={kb} VAR={var}: logits rel err {np.abs(lg - want).max() / np.abs(want).max():.3e}")
