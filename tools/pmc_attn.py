#!/usr/bin/env python3
"""SQ counters of the attention kernels (forward 8-wave / 4-wave, backward dq / dkdv) at the cfg2 shape.
   rocprofv3 --pmc <8 SQ counters> --kernel-trace --output-format csv -d <dir> -- python3 tools/pmc_attn.py run
   python3 tools/pmc_attn.py sum <dir>... <out.json>"""
import collections, csv, glob, json, os, sys


def run():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from gaviko_amd import lib, ops
    lib.require_device()
    dev = torch.device("cuda:0")
    B, T, H = 4, 1033, 12
    inner = H * 64
    qkv = ops.act_zeros(B * T, 3 * inner, torch.bfloat16, dev); qkv[:B * T] = (torch.randn(B * T, 3 * inner, device=dev) * 0.7).bfloat16()
    out, dout = ops.act_zeros(B * T, inner, torch.bfloat16, dev), ops.act_zeros(B * T, inner, torch.bfloat16, dev)
    dout[:B * T] = torch.randn(B * T, inner, device=dev).bfloat16()
    lse, delta = torch.empty(B * H * T, device=dev), torch.empty(B * H * T, device=dev)
    dqkv = ops.act_zeros(B * T, 3 * inner, torch.bfloat16, dev)
    ops.qkv_prescale(qkv, B * T, H, 0.125)
    wsp = ops.attention_bwd_workspace(B, T, H, dev)
    for _ in range(6):
        ops.attention_fwd(qkv, out, lse, B, T, H, 0.125, q_prescaled=True)
        ops.attention_bwd(qkv, out, dout, lse, delta, dqkv, B, T, H, 0.125, q_prescaled=True)                 # two passes
        ops.attention_bwd(qkv, out, dout, lse, delta, dqkv, B, T, H, 0.125, q_prescaled=True, ws=wsp)         # one pass
    torch.cuda.synchronize()


def summarise(dirs, out_path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for d in dirs:
        f = max(glob.glob(f"{d}/*/*_counter_collection.csv"), key=os.path.getmtime)
        per = collections.defaultdict(dict)
        for r in csv.DictReader(open(f)):
            if "attn" not in r["Kernel_Name"]:
                continue
            k = (r["Kernel_Name"].split("(")[0].replace("void gvk::", ""), int(r["Dispatch_Id"]))
            per[k][r["Counter_Name"]] = per[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        seen = collections.Counter()
        for (name, did), c in sorted(per.items(), key=lambda kv: kv[0][1]):
            seen[name] += 1
            if seen[name] <= 2:
                continue                                   # skip cold launches
            for cn, v in c.items():
                acc[name][cn].append(v)
        for kt in glob.glob(f"{os.path.dirname(f)}/*_kernel_trace.csv"):
            for r in csv.DictReader(open(kt)):
                if "attn" in r["Kernel_Name"]:
                    dur[r["Kernel_Name"].split("(")[0].replace("void gvk::", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    out = {}
    for name, cs in acc.items():
        c = {k: sum(v) / len(v) for k, v in cs.items()}
        o = {"counters": c, "us_profiled": round(sum(dur[name]) / max(1, len(dur[name])), 1)}
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            for a, nm in (("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_WAIT_INST_ANY", "wait_inst_any_frac"), ("SQ_WAIT_INST_LDS", "lds_issue_stall_frac"),
                          ("SQ_ACTIVE_INST_VALU", "valu_active_frac"), ("SQ_ACTIVE_INST_LDS", "lds_active_frac"), ("SQ_ACTIVE_INST_ANY", "active_frac")):
                if a in c:
                    o[nm] = round(c[a] / wc, 4)
        if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
            o["lds_bank_conflict_frac"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
        out[name] = o
        print(name, json.dumps({k: v for k, v in o.items() if k != "counters"}), {k: round(v) for k, v in c.items()})
    json.dump(out, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else summarise(sys.argv[2:-1], sys.argv[-1])
