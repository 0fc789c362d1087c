set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/dy16; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_sidepath_kernels_gpu.py tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q > $O/test.log 2>&1 || { tail -30 $O/test.log; exit 1; }
tail -2 $O/test.log
run() { echo -n "$1: "; shift; env GAVIKO_HIP_DIAG=1 "$@" python bench.py --allow-diag --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*'; }
for i in 1 2 3; do
  run dy16 GAVIKO_HIP_DY16=1
  run dy32 GAVIKO_HIP_DY16=0
done | tee $O/ab.txt
env GAVIKO_HIP_DIAG=1 timeout -k 10 200 python tools/plan_marks.py 4 vit-b16 > $O/marks.txt 2>&1
