#!/usr/bin/env python3
"""SQ counters of the bf16 GEMM kernels on the hot-path shapes (MFMA utilisation, LDS stalls, bank conflicts).

GPU box, two modes:
  python3 tools/pmc_gemm.py run [tiles]        the workload: every hot-path shape x tile, ITERS launches each (what rocprofv3 wraps)
  python3 tools/pmc_gemm.py sum <dir>... <out.json>   merge the counter CSVs of one or more `rocprofv3 --pmc` passes into per-(shape, tile) means

  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
            SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_gemm_a -- python3 tools/pmc_gemm.py run
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts
cycles (16 per v_mfma_f32_16x16x32_bf16, 32 per 32x32x16) summed over SIMDs; SQ_BUSY_CYCLES is per SE.  MFMA utilisation of a launch is
reported as MFMA_BUSY / (4 SIMDs x 256 CUs x GRBM-free estimate) AND, independent of the counter's scaling, as achieved TFLOP/s from the
kernel-trace duration of the same dispatch.
"""
import collections
import csv
import glob
import json
import os
import sys

SHAPES = [("qkv", 2304, 768, "store_bf16"), ("out", 768, 768, "bias_res_f32"), ("fc1", 3072, 768, "bias_gelu_bf16"), ("fc2", 768, 3072, "bias_res_f32"),
          ("fc2_dgrad", 3072, 768, "gelu_bwd_bf16"), ("fc1_dgrad", 768, 3072, "store_f32"), ("out_dgrad", 768, 768, "store_bf16"),
          ("qkv_dgrad", 768, 2304, "store_f32")]
M = 4132


def run(tiles):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from gaviko_amd import lib, ops
    lib.require_device()
    dev = torch.device("cuda:0")
    epi = dict(store_bf16=ops.EPI_STORE_BF16, bias_res_f32=ops.EPI_BIAS_RES_F32, bias_gelu_bf16=ops.EPI_BIAS_GELU_BF16,
               gelu_bwd_bf16=ops.EPI_GELU_BWD_BF16, store_f32=ops.EPI_STORE_F32)
    iters = int(os.environ.get("ITERS", "6"))
    order = []
    for name, N, K, e in SHAPES:
        a = ops.act_zeros(M, K, torch.bfloat16, dev); a[:M] = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
        f32 = e in ("bias_res_f32", "store_f32")
        out0 = ops.act_zeros(M, N, torch.float32 if f32 else torch.bfloat16, dev)
        out1 = ops.act_zeros(M, N, torch.bfloat16, dev)
        bias = torch.randn(N, device=dev)
        res = ops.act_zeros(M, N, torch.float32, dev)
        aux = ops.act_zeros(M, N, torch.bfloat16, dev); aux.normal_()
        torch.cuda.synchronize()
        for t in tiles:
            kw = dict(epilogue=epi[e], tile=t)
            if e == "bias_res_f32": kw.update(bias=bias, res=res)
            if e == "bias_gelu_bf16": kw.update(bias=bias, out1=out1)
            if e == "gelu_bwd_bf16": kw.update(aux=aux)
            try:
                for _ in range(iters):
                    ops.gemm_nt(a, w, M, out0, **kw)
            except Exception as ex:                       # a tile that is not built for this epilogue
                print(f"skip {name} tile {t}: {ex}", flush=True)
                continue
            order.append((name, N, K, e, t, iters))
        torch.cuda.synchronize()
    # the launch order is what `sum` uses to attribute dispatches: GEMM dispatches appear in exactly this order
    json.dump(order, open(os.environ.get("PMC_GEMM_ORDER", "gpurun_out/pmc_gemm_order.json"), "w"))
    print(f"{len(order)} (shape, tile) cases x {iters} launches", flush=True)


def summarise(dirs, out_path):
    order = json.load(open(os.environ.get("PMC_GEMM_ORDER", "gpurun_out/pmc_gemm_order.json")))
    cases = [dict(shape=n, M=M, N=N, K=K, epilogue=e, tile=t, flops=2.0 * M * N * K, counters={}) for n, N, K, e, t, _ in order]
    per = [it for *_, it in order]
    for d in dirs:
        f = max(glob.glob(f"{d}/*/*_counter_collection.csv"), key=os.path.getmtime)
        byd = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            if "gemm" not in r["Kernel_Name"]:
                continue
            rec = byd.setdefault(int(r["Dispatch_Id"]), {"kernel": r["Kernel_Name"], "c": {}, "vgpr": r.get("VGPR_Count"), "lds": r.get("LDS_Block_Size"),
                                                         "grid": r.get("Grid_Size"), "wg": r.get("Workgroup_Size")})
            rec["c"][r["Counter_Name"]] = rec["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        disp = [byd[k] for k in sorted(byd)]
        assert len(disp) == sum(per), f"{d}: {len(disp)} GEMM dispatches, expected {sum(per)}"
        # durations of the same dispatches from the kernel trace of this pass
        kt = glob.glob(f"{os.path.dirname(f)}/*_kernel_trace.csv")
        dur = {}
        if kt:
            for r in csv.DictReader(open(kt[0])):
                if "gemm" in r["Kernel_Name"]:
                    dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
        ids = sorted(byd)
        pos = 0
        for c, n in zip(cases, per):
            chunk, cid = disp[pos:pos + n], ids[pos:pos + n]
            pos += n
            c["kernel"] = chunk[0]["kernel"].replace("void ", "").replace("gvk::", "")
            c.update(vgpr=chunk[0]["vgpr"], lds_bytes=chunk[0]["lds"], workgroups=int(chunk[0]["grid"]) // max(1, int(chunk[0]["wg"])), waves_per_wg=int(chunk[0]["wg"]) // 64)
            use = chunk[1:] if n > 1 else chunk            # drop the first (cold) launch
            for name in use[0]["c"]:
                c["counters"][name] = sum(x["c"][name] for x in use) / len(use)
            ds = [dur[i] for i in cid[1:] if i in dur]
            if ds:
                c.setdefault("us_profiled", []).append(round(sum(ds) / len(ds), 2))
    for c in cases:
        k = c["counters"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in k and "SQ_BUSY_CYCLES" in k and k["SQ_BUSY_CYCLES"] > 0:
            # guide: MFMA_BUSY = cycles the matrix pipe is busy, summed over SIMDs; the wall-clock cycles of the launch are taken from
            # its profiled duration x an assumed 2.0 GHz only as a cross-check -- the ratio below uses counters alone
            pass
        if "us_profiled" in c:
            us = sum(c["us_profiled"]) / len(c["us_profiled"])
            c["us_profiled"] = round(us, 2)
            c["tflops_profiled"] = round(c["flops"] / us / 1e6, 1)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in k:
                nmfma_cycles = c["flops"] / (2 * 16 * 16 * 32) * 16          # 16 cycles per 16x16x32 MFMA
                c["mfma_busy_per_simd_cycles"] = round(k["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024, 1)       # 256 CUs x 4 SIMDs
                c["mfma_ideal_per_simd_cycles"] = round(nmfma_cycles / 1024, 1)
                c["mfma_util_at_2p4GHz"] = round(k["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (us * 2400.0), 3)
        for a, b, name in (("SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES", "lds_issue_stall_frac"), ("SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "wait_any_frac"),
                           ("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "wait_inst_any_frac"), ("SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES", "active_inst_frac"),
                           ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "lds_bank_conflict_frac"), ("SQ_ACTIVE_INST_LDS", "SQ_WAVE_CYCLES", "lds_inst_frac"),
                           ("SQ_INST_CYCLES_VMEM", "SQ_WAVE_CYCLES", "vmem_inst_frac")):
            if a in k and b in k and k[b] > 0:
                c[name] = round(k[a] / k[b], 4)
    json.dump({"note": __doc__.split("Units")[1].strip(), "M": M, "cases": cases}, open(out_path, "w"), indent=1)
    for c in cases:
        print(f"{c['shape']:10s} tile {c['tile']:8d} {c.get('us_profiled', 0):7.1f} us {c.get('tflops_profiled', 0):7.1f} TF  mfma_util {c.get('mfma_util_at_2p4GHz', 0):.3f} "
              f"wait_any {c.get('wait_any_frac', 0):.3f} wait_inst {c.get('wait_inst_any_frac', 0):.3f} lds_stall {c.get('lds_issue_stall_frac', 0):.3f} "
              f"bank_conf {c.get('lds_bank_conflict_frac', 0):.3f}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run([int(t) for t in sys.argv[2].split(",")] if len(sys.argv) > 2 else [8256256, 256256, 3128128, 128128])
    else:
        summarise(sys.argv[2:-1], sys.argv[-1])
