# in-process A/B of the 128-row halo kernel's tap order (dev knob 0: 5 = round 4, every pixel fragment read twice per tap; 0 = shipped MF 21, once)
set -e
for cfg in "256 128 128" "128 256 256" "64 512 512"; do
  set -- $cfg
  python scripts/bench_layer.py --res $1 --cin $2 --cout $3 --batch 8 --prec 3 --ab 0 --va 5 --vb 0
  python scripts/bench_layer.py --res $1 --cin $2 --cout $3 --batch 8 --prec 3 --bwd --ab 0 --va 5 --vb 0
done
