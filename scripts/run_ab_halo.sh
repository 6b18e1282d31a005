#!/bin/bash
# in-process A/B of a dev knob on the three stride-1 probe layers (154.6 GFLOP each), forward and backward
# usage: run_ab_halo.sh KNOB [VA VB]
K=${1:-0}; VA=${2:-0}; VB=${3:-1}
for cfg in "256 128 128" "128 256 256" "64 512 512"; do
  set -- $cfg
  python scripts/bench_layer.py --res $1 --cin $2 --cout $3 --batch 8 --prec 3 --ab $K --va $VA --vb $VB
  python scripts/bench_layer.py --res $1 --cin $2 --cout $3 --batch 8 --prec 3 --bwd --ab $K --va $VA --vb $VB
done
