#!/bin/bash
# round 3, call k: SQ counters of the new attention kernels (two PMC passes, counters + kernel trace only)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r3k
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/r3k/a -- python3 tools/pmc_attn.py run > gpurun_out/r3k/a.log 2>&1; echo "pass a rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU --kernel-trace --output-format csv -d gpurun_out/r3k/b -- python3 tools/pmc_attn.py run > gpurun_out/r3k/b.log 2>&1; echo "pass b rc=$?"
python3 tools/pmc_attn.py sum gpurun_out/r3k/a gpurun_out/r3k/b gpurun_out/r3k/pmc_attention.json; echo "sum rc=$?"
rm -rf gpurun_out/r3k/a gpurun_out/r3k/b
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r3k/pmc_attention.json"))
for k,v in d.items():
    if isinstance(v,dict):
        for kn,kv in v.items():
            if isinstance(kv,dict) and "counters" in kv:
                c=kv["counters"]; print(kn, {x:kv[x] for x in kv if x!="counters"}); print("   ", {x:int(y) for x,y in c.items()})
PY
