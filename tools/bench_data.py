"""GPU box: the data-side kernels (csrc/augment.hip) on a batch of 4 raw (120,160,160) volumes, timed by replaying 50 recorded launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gaviko_amd import data, lib, ops
lib.require_device()
dev = torch.device("cuda:0")
B, shape = 4, (120, 160, 160)
V = int(np.prod(shape))
x = (torch.randn(B, 1, *shape, device=dev) * 300 + 1000).contiguous()
out = torch.empty_like(x)
part = ops.minmax_partials(B, dev)
mats = np.stack([data.affine_matrix((1.05, 0.95, 1.02), (10, -12, 14), (0, 0, 0), shape).astype(np.float32)] * B)
mats_d = torch.from_numpy(mats).to(dev)
fl_aff = torch.full((B,), 9, dtype=torch.int32, device=dev)
fl_flip = torch.full((B,), 1, dtype=torch.int32, device=dev)


def t(name, fn, nbytes):
    for _ in range(3): fn()
    l = lib.load()
    lib.check(l.gvk_plan_begin(), "begin")
    for _ in range(50): fn()
    pid = l.gvk_plan_end()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    l.gvk_plan_replay(pid)
    e0.record(); l.gvk_plan_replay(pid); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"{name:34s} {us:7.1f} us  {nbytes / us / 1e6:6.2f} TB/s algorithmic ({nbytes / 1e6:.1f} MB)")
    l.gvk_plan_free(pid)


t("volume_minmax", lambda: ops.volume_minmax(x, part), B * V * 4)
t("rescale_intensity", lambda: ops.rescale_intensity(x, part, out), 2 * B * V * 4)
t("spatial: flip axis 0", lambda: ops.spatial_transform(x, out, mats_d, fl_flip, part), 2 * B * V * 4)
t("spatial: affine + flip (trilinear)", lambda: ops.spatial_transform(x, out, mats_d, fl_aff, part), 2 * B * V * 4)
