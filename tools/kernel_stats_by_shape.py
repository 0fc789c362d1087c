#!/usr/bin/env python3
"""Per-SHAPE kernel statistics from a `rocprofv3 --kernel-trace` CSV.

`rocprofv3 --stats` groups by kernel NAME only, and one GEMM instantiation serves several shapes of a step (the three-stage
128x128 `gemm_nt_kernel<128,128,1,...>` runs both the out-projection, K = 768, and fc2 forward, K = 3136 -- same name AND same grid),
so its "average duration" is a mixture nobody can price.  This groups the dispatches of our kernels by

    (kernel name, grid size, workgroup size, name of the PREVIOUS kernel on the same stream)

-- a step is a fixed launch sequence per stream, so the predecessor identifies the call site (fc2 forward always follows the fc1
GEMM, the out-projection always follows the attention forward, ...) -- and labels the GEMM call sites of the GAViKO step with their
M/N/K.  The rows are what `bench.py`'s `roofline` / `gemm_classes` figures are to be checked against.

usage: python tools/kernel_stats_by_shape.py <dir or kernel_trace.csv> [--out profiles/r03_kernel_stats_by_shape.csv] [--skip-first N]
"""
import argparse
import collections
import csv
import glob
import os
import re
import statistics
import sys


def short(name: str) -> str:
    mm = re.match(r"_ZN3gvk(\d+)", name)
    if mm:
        name = name[mm.end(): mm.end() + int(mm.group(1))]
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "").replace("gvk::", "")
    depth, out = 0, []
    for ch in name:                                   # drop the argument list, keep the template arguments
        if ch == "(" and depth == 0:
            break
        depth += ch == "<"
        depth -= ch == ">"
        out.append(ch)
    return "".join(out).strip()


# GEMM call sites of the gaviko step: (kernel-name regex, predecessor regex) -> label.  M/N/K are filled from --M/--C/--mlp.
def gemm_labels(M, C, mlp, ldx):
    return [
        (r"gemm8p_kernel<0,", r".*", f"qkv fwd        M={M} N={3 * C} K={C}"),
        (r"gemm_nt_kernel<\d+, \d+, 1,", r"attn_fwd", f"out-proj fwd   M={M} N={C} K={C}"),
        (r"gemm8p_kernel<2,", r".*", f"fc1 fwd        M={M} N={mlp} K={C}"),
        (r"gemm_nt_kernel<\d+, \d+, [16],", r"gemm8p_kernel<2,", f"fc2 fwd (+GPA up-projection)  M={M} N={C} K={ldx} (algorithmic K {mlp}+20)"),
        (r"gemm8p_kernel<4,", r".*", f"fc2 dgrad      M={M} N={mlp} K={C}"),
        # (epilogue 5 = fp32 store; 0 = bf16 store since round 5 hands the LayerNorm backward a bf16 gradient: the predecessor tells the sites apart;
        #  the panel launches of the pruned top / bottom layers follow another 64 x 128 panel GEMM or the dK/dV pass)
        (r"gemm_nt_kernel<\d+, \d+, [05],", r"gemm8p_kernel<4,|gemm_nt_kernel<64, 128, 4,", f"fc1 dgrad      M={M} N={C} K={mlp}"),
        (r"gemm_nt_kernel<\d+, \d+, [05],", r"attn_bwd", f"qkv dgrad      M={M} N={C} K={3 * C}"),
        (r"gemm_nt_kernel<\d+, \d+, 0,", r".*", f"out-proj dgrad M={M} N={C} K={C}"),
        (r"gemm_nt_kernel<\d+, \d+, 3,", r"patchify", f"patch embed    M={M // 1033 * 1000 if M % 1033 == 0 else '?'} N={C} K=3072"),
    ]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("--out", default=None)
    ap.add_argument("--skip-first", type=int, default=0, help="ignore the first N dispatches of every group (warm-up / eager passes)")
    ap.add_argument("--M", type=int, default=4132)
    ap.add_argument("--C", type=int, default=768)
    ap.add_argument("--mlp", type=int, default=3072)
    a = ap.parse_args()
    f = a.src
    if os.path.isdir(f):
        cands = glob.glob(os.path.join(f, "**", "*kernel_trace.csv"), recursive=True)
        if not cands:
            sys.exit(f"no *kernel_trace.csv under {f}")
        f = max(cands, key=os.path.getsize)
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        r["n"] = short(r["Kernel_Name"])
    rows.sort(key=lambda r: r["s"])
    skey = "Stream_Id" if "Stream_Id" in rows[0] and len({r["Stream_Id"] for r in rows}) > 1 else "Queue_Id"
    prev = {}
    groups = collections.defaultdict(list)
    for r in rows:
        st = r[skey]
        p = prev.get(st, "-")
        prev[st] = r["n"]
        if "at::native" in r["n"] or r["n"].startswith(("Cijk", "rccl", "nccl")):
            continue
        grid = 1
        for ax in "XYZ":
            grid *= max(1, int(r[f"Grid_Size_{ax}"]) // max(1, int(r[f"Workgroup_Size_{ax}"])))
        groups[(r["n"], grid, int(r["Workgroup_Size_X"]), p)].append((r["e"] - r["s"]) / 1e3)
    labels = gemm_labels(a.M, a.C, a.mlp, a.mlp + 64)
    out = []
    for (n, grid, wg, p), d in groups.items():
        d = d[a.skip_first:] if len(d) > a.skip_first + 4 else d
        lab = ""
        for kre, pre, text in labels:
            if re.match(kre, n) and re.match(pre, p):
                lab = text
                break
        out.append(dict(kernel=n, workgroups=grid, wg_size=wg, after=p, calls=len(d), avg_us=round(statistics.fmean(d), 2),
                        median_us=round(statistics.median(d), 2), min_us=round(min(d), 2), max_us=round(max(d), 2),
                        total_ms=round(sum(d) / 1e3, 3), call_site=lab))
    out.sort(key=lambda r: -r["total_ms"])
    w = csv.DictWriter(open(a.out, "w", newline="") if a.out else sys.stdout, fieldnames=list(out[0].keys()))
    w.writeheader()
    w.writerows(out)
    if a.out:
        print(f"{len(out)} (kernel, grid, predecessor) groups from {len(rows)} dispatches of {f} ({skey}) -> {a.out}", file=sys.stderr)


if __name__ == "__main__":
    main()
