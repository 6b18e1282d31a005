#!/bin/bash
# in-process A/B of a dev knob on the up-sampling layers (flat kernel: transposed-conv phases forward, stride-2 backward) and on the
# split-K sizes: run_ab_up.sh KNOB VA VB
K=${1:-2}; VA=${2:-8}; VB=${3:-0}
for cfg in "256 256 128" "128 512 256" "64 512 512" "32 512 512" "16 512 512"; do
  set -- $cfg
  python scripts/bench_layer.py --res $1 --cin $2 --cout $3 --batch 8 --prec 3 --up --ab $K --va $VA --vb $VB --rounds 5
  python scripts/bench_layer.py --res $1 --cin $2 --cout $3 --batch 8 --prec 3 --up --bwd --ab $K --va $VA --vb $VB --rounds 5
done
for cfg in "32 512 512" "16 512 512"; do
  set -- $cfg
  python scripts/bench_layer.py --res $1 --cin $2 --cout $3 --batch 8 --prec 3 --ab $K --va $VA --vb $VB --rounds 5 --iters 50
  python scripts/bench_layer.py --res $1 --cin $2 --cout $3 --batch 8 --prec 3 --bwd --ab $K --va $VA --vb $VB --rounds 5 --iters 50
done
