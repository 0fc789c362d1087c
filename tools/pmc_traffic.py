#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- separate runs, as the MI355X guide prescribes) of `bench.py` into
profiles/<tag>_pmc_traffic.json: HBM-side bytes per launch for every GEMM instantiation and the other heavy kernels.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...   (same command)
  python bench.py --steps 2 --warmup 2 --no-cpu-baseline --dump-gemm-shapes gpurun_out/gemm_shapes.json      (same workload flags)
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r05_pmc_traffic.json gpurun_out/gemm_shapes.json

The profiler names kernel INSTANTIATIONS; one instantiation serves several GEMM shapes (fc1 dgrad K = 3072 and qkv dgrad K = 2304 share
the plain-store kernel) and the same instantiation serves other shapes at another batch / backbone.  The fourth argument (bench.py's
own list of this workload's GEMM classes) lets every traffic cluster be stored WITH the [M, N, K] it was measured on; bench.py
attaches a figure to a roofline line only when that shape equals the line's.

gfx950 corrections (MI355X_MICROARCH.md, HBM): both counters are in KiB; FETCH_SIZE reports exactly half the bytes of a wide
coalesced read stream, so reads = 2 * FETCH_SIZE * 1024.  Calibrated here on patchify_kernel, whose traffic is known exactly
(49.152 MB read, 24.576 MB written per 4-volume launch).
"""
import collections
import csv
import glob
import json
import sys


def load(d, counter):
    """-> {kernel: [per-dispatch values in dispatch order]}"""
    import os
    f = max(glob.glob(f"{d}/*/*_counter_collection.csv"), key=os.path.getmtime)
    acc = collections.defaultdict(list)
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def clusters(reads, writes):
    """One kernel instantiation may serve several shapes (e.g. the bias+residual GEMM: out-proj K=768 and fc2 K=3072): split its
    dispatches into groups at every gap of more than 15 % between consecutive sorted read figures (fc1 dgrad K=3072 and qkv dgrad K=2304 share
    the plain-store instantiation and differ by 1.3x; launches of one shape repeat within a few per cent)."""
    order = sorted(range(len(reads)), key=lambda i: reads[i])
    groups = []
    for i in order:
        if groups and reads[i] <= 1.15 * groups[-1][-1][0] + 1:
            groups[-1].append((reads[i], writes[i] if i < len(writes) else 0.0))
        else:
            groups.append([(reads[i], writes[i] if i < len(writes) else 0.0)])
    out = []
    for g in groups:
        rd, wr = sum(x[0] for x in g) / len(g), sum(x[1] for x in g) / len(g)
        out.append({"launches": len(g), "read_bytes": round(2 * rd * 1024), "write_bytes": round(wr * 1024), "total_bytes": round(2 * rd * 1024 + wr * 1024)})
    return out


def kernel_family(kernel_name):
    """Which launches a GEMM instantiation serves: 'wide' = the eight-phase / 256 x 256 kernels (N >= 2304 shapes), 'panels' = the
    four-stage 64 x 128 tile of the strided row-panel launches (dead-row pruning), 'tile' = every other 4-wave tile."""
    import re
    if kernel_name.startswith("gemm8p_kernel") or re.match(r"gemm_nt_kernel<256, 256,", kernel_name):
        return "wide"
    if re.match(r"gemm_nt_kernel<64, 128, \d+, 64, false, 4, 4", kernel_name):
        return "panels"
    return "tile"


def class_family(class_name, shape):
    """The same split on bench.py's side (engine.py / gemm_bf16.hip::dispatch_tile): classes named '... panels=BxR' are panel launches;
    N % 256 == 0 shapes with 140..256 tiles of 256 x 256 run on the eight-phase kernel."""
    if "panels=" in class_name:
        return "panels"
    M, N, K = shape
    t256 = ((M + 255) // 256) * (N // 256) if N % 256 == 0 else 0
    return "wide" if 140 <= t256 <= 256 else "tile"


def attach_shapes(kernels, shapes_doc, epi_of, alg_bytes, epi_names):
    """Give every cluster of a GEMM instantiation the [M, N, K] (and class name) it was measured on: the workload's classes with that
    epilogue AND kernel family, in ascending algorithmic bytes, against the clusters in ascending traffic -- only when the counts agree
    and no cluster moved fewer bytes than its shape needs (then the pairing cannot be right and the clusters stay unlabelled)."""
    for k, v in kernels.items():
        epi = epi_of(k)
        if epi is None or epi not in epi_names:
            continue
        fam = kernel_family(k)
        cand = sorted((c for c in shapes_doc["classes"] if c.startswith(f"gemm_nt_bf16[{epi_names[epi]}]")
                       and class_family(c, shapes_doc["classes"][c]) == fam),
                      key=lambda c: alg_bytes(epi, shapes_doc["classes"][c]))
        cl = v.get("clusters") or []
        if not cand or len(cand) != len(cl):
            continue
        if any(c["total_bytes"] < 0.9 * alg_bytes(epi, shapes_doc["classes"][n]) for c, n in zip(cl, cand)):
            continue
        for c, n in zip(cl, cand):
            c["shape"] = list(shapes_doc["classes"][n])
            c["class"] = n
            c["algorithmic_bytes"] = alg_bytes(epi, c["shape"])


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
    import bench
    out = {"gemm_source_sha": bench.gemm_source_hash(), "note": "bytes per launch; read = 2*FETCH_SIZE*1024 (gfx950 half-count correction), write = WRITE_SIZE*1024; 'clusters' splits "
                   "an instantiation that serves several shapes (ascending traffic)", "kernels": {}}
    for k in sorted(fetch, key=lambda k: -sum(fetch[k])):
        short = k.replace("gvk::", "").replace("void ", "")
        n = len(fetch[k])
        rd, wr = 2 * sum(fetch[k]) / n * 1024, (sum(write.get(k, [0.0])) / max(1, len(write.get(k, [0.0])))) * 1024
        out["kernels"][short] = {"launches": n, "read_bytes": round(rd), "write_bytes": round(wr), "total_bytes": round(rd + wr),
                                 "clusters": clusters(fetch[k], write.get(k, []))}
        if "gemm" in short:                     # the per-launch figures the clusters were cut from (KiB as counted, sorted)
            out["kernels"][short]["fetch_kib_sorted"] = [round(x) for x in sorted(fetch[k])]
    if len(sys.argv) > 4:
        from gaviko_amd.engine_common import _EPI_NAMES
        shapes_doc = json.load(open(sys.argv[4]))
        out["workload"] = shapes_doc["workload"]
        attach_shapes(out["kernels"], shapes_doc, bench.gemm_kernel_epilogue, bench.gemm_alg_bytes, _EPI_NAMES)
    pk = [k for k in out["kernels"] if k.startswith("patchify_kernel")]
    if pk:
        out["calibration"] = {"kernel": pk[0], "expected_read": 4 * 120 * 160 * 160 * 4, "expected_write": 4 * 1000 * 3072 * 2, **out["kernels"][pk[0]]}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(f"wrote {sys.argv[3]}: {len(out['kernels'])} kernels")


if __name__ == "__main__":
    main()
