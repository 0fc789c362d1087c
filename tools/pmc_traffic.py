#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- separate runs, as the MI355X guide prescribes) of `bench.py` into
profiles/<tag>_pmc_traffic.json: HBM-side bytes per launch for every GEMM instantiation and the other heavy kernels.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...   (same command)
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json

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
    f = glob.glob(f"{d}/*/*_counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    fetch, n = load(sys.argv[1], "FETCH_SIZE")
    write, _ = load(sys.argv[2], "WRITE_SIZE")
    out = {"note": "bytes per launch; read = 2*FETCH_SIZE*1024 (gfx950 half-count correction), write = WRITE_SIZE*1024", "kernels": {}}
    for k in sorted(fetch, key=lambda k: -fetch[k] * n[k]):
        short = k.replace("gvk::", "").replace("void ", "")
        rd, wr = 2 * fetch[k] * 1024, write.get(k, 0.0) * 1024
        out["kernels"][short] = {"launches": n[k], "read_bytes": round(rd), "write_bytes": round(wr), "total_bytes": round(rd + wr)}
    pk = [k for k in out["kernels"] if k.startswith("patchify_kernel")]
    if pk:
        out["calibration"] = {"kernel": pk[0], "expected_read": 4 * 120 * 160 * 160 * 4, "expected_write": 4 * 1000 * 3072 * 2, **out["kernels"][pk[0]]}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(f"wrote {sys.argv[3]}: {len(out['kernels'])} kernels")


if __name__ == "__main__":
    main()
