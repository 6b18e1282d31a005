#!/bin/bash
# several A/B pairs of one dev knob on the probe layers: run_ab_pairs.sh KNOB "VA:VB VA:VB ..."
K=${1:-0}; shift
for pair in $@; do
  VA=${pair%%:*}; VB=${pair##*:}
  for cfg in "256 128 128" "128 256 256" "64 512 512"; do
    set -- $cfg
    python scripts/bench_layer.py --res $1 --cin $2 --cout $3 --batch 8 --prec 3 --ab $K --va $VA --vb $VB --rounds 5
    python scripts/bench_layer.py --res $1 --cin $2 --cout $3 --batch 8 --prec 3 --bwd --ab $K --va $VA --vb $VB --rounds 5
  done
done
