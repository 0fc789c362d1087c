#!/usr/bin/env python3
"""Timeline analysis of a rocprofv3 --kernel-trace CSV: isolates the last replayed training step, groups dispatches per queue,
and prints per-queue busy time, idle gaps of the busiest (main) queue and the kernels that ran before the longest gaps.
usage: python tools/timeline.py gpurun_out/ktrace [ngaps]"""
import collections
import csv
import glob
import re
import sys

import os
f = max(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(f))]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"]
    mm = re.match(r"_ZN3gvk(\d+)", name)
    if mm:                                                               # un-demangled template instance: keep the function name
        name = name[mm.end():mm.end() + int(mm.group(1))]
    r["n"] = re.sub(r"\(.*", "", name.replace("gvk::", "").replace("void ", ""))[:44]
rows.sort(key=lambda r: r["s"])
# last step = from the last patchify launch to the end
starts = [i for i, r in enumerate(rows) if r["n"].startswith("patchify")]
step = rows[starts[-1]:]
t0, t1 = step[0]["s"], max(r["e"] for r in step)
print(f"step span {(t1 - t0) / 1e3:.1f} us, {len(step)} dispatches")
byq = collections.defaultdict(list)
for r in step:
    byq[r["Queue_Id"]].append(r)
for q, rs in byq.items():
    busy = sum(r["e"] - r["s"] for r in rs)
    print(f"queue {q}: {len(rs)} kernels, busy {busy / 1e3:.1f} us, first {(rs[0]['s'] - t0) / 1e3:.1f} last {(rs[-1]['e'] - t0) / 1e3:.1f}")
# the backbone chain by kernel name (graph replays map capture streams to hardware queues freely, so queue ids do not identify it)
main = [r for r in step if r["n"].startswith(("gemm_nt", "gemm8p", "attn_", "ln_fwd"))]
gaps = []
for a, b in zip(main, main[1:]):
    gaps.append((b["s"] - a["e"], a, b))
tot = sum(g[0] for g in gaps)
print(f"main queue: busy {sum(r['e'] - r['s'] for r in main) / 1e3:.1f} us, gaps {tot / 1e3:.1f} us over {len(gaps)}")
hist = collections.Counter()
for g, a, b in gaps:
    hist[(a["n"], b["n"])] += g
print("gap time by (prev -> next) kernel pair on the main queue:")
for (a, b), g in hist.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 25):
    cnt = sum(1 for gg, aa, bb in gaps if aa["n"] == a and bb["n"] == b)
    print(f"  {g / 1e3:8.1f} us  n={cnt:3d}  {a}  ->  {b}")
# per-kernel average inside this step
agg = collections.defaultdict(lambda: [0, 0])
for r in step:
    agg[(r["Queue_Id"], r["n"])][0] += r["e"] - r["s"]
    agg[(r["Queue_Id"], r["n"])][1] += 1
print("kernel time by queue:")
for (q, n), (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"  q{q} {n:46s} {c:4d} x {t / c / 1e3:7.1f} us = {t / 1e3:8.1f}")

if len(sys.argv) > 3:   # dump a window: every dispatch between the k-th and (k+1)-th fc2-dgrad GEMM of the step
    k = int(sys.argv[3])
    marks = [r for r in step if r["n"].startswith("gemm8p_kernel<4")]
    a, b = marks[k]["s"], marks[k + 1]["e"]
    print(f"window between fc2-dgrad #{k} and #{k + 1}: {(b - a) / 1e3:.1f} us")
    for r in step:
        if r["e"] > a and r["s"] < b:
            print(f"  q{r['Queue_Id']} {(r['s'] - a) / 1e3:8.1f} -> {(r['e'] - a) / 1e3:8.1f}  ({(r['e'] - r['s']) / 1e3:6.1f})  {r['n']}")
