#!/bin/bash
# in-process A/B of a dev knob on the small-grid layers (4^2 .. 32^2; stride-1, up forward, up backward): run_ab_small.sh KNOB VA VB
K=${1:-3}; VA=${2:-8}; VB=${3:-0}
for res in 4 8 16 32; do
  python scripts/bench_layer.py --res $res --cin 512 --cout 512 --batch 8 --prec 3 --ab $K --va $VA --vb $VB --rounds 5 --iters 50
  python scripts/bench_layer.py --res $res --cin 512 --cout 512 --batch 8 --prec 3 --bwd --ab $K --va $VA --vb $VB --rounds 5 --iters 50
done
for res in 8 16 32 64; do
  python scripts/bench_layer.py --res $res --cin 512 --cout 512 --batch 8 --prec 3 --up --ab $K --va $VA --vb $VB --rounds 5 --iters 50
  python scripts/bench_layer.py --res $res --cin 512 --cout 512 --batch 8 --prec 3 --up --bwd --ab $K --va $VA --vb $VB --rounds 5 --iters 50
done
