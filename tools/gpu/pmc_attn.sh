set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5i; mkdir -p $O
A1="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"
A2="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc $A1 --kernel-trace --output-format csv -d $O/pa_a -- python3 $R/tools/pmc_attn.py run > $O/pa_a.log 2>&1
timeout -k 10 300 rocprofv3 --pmc $A2 --kernel-trace --output-format csv -d $O/pa_b -- python3 $R/tools/pmc_attn.py run > $O/pa_b.log 2>&1
python3 $R/tools/pmc_attn.py sum $O/pa_a $O/pa_b $O/pmc_attention.json > $O/pa_sum.log 2>&1
rm -rf $O/pa_a $O/pa_b
cat $O/pa_sum.log
