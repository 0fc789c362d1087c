#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "gemm or gaviko or golden" 2>&1 | tail -2
run() { echo -n "$1: "; python bench.py --steps 30 --warmup 10 $2 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|rror.*' ; }
run "ViT-B B=4" ""
run "ViT-B B=2" "--batch 2"
run "ViT-B B=1" "--batch 1"
run "ViT-B B=8" "--batch 8"
run "ViT-L B=2" "--backbone vit-l16 --batch 2"
run "ViT-T B=4" "--backbone vit-t16 --batch 4"
