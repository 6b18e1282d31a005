# dev tool: forced K-slice counts (knob 3) against the cost model's choice (0) on the split-K layers:  run_ab_ksplit.sh
for res in 32; do for v in 3 4; do for bw in "" "--bwd"; do
  python scripts/bench_layer.py --res $res --cin 512 --cout 512 --batch 8 --prec 3 $bw --ab 3 --va 0 --vb $v --rounds 5 --iters 50 2>/dev/null | cut -c1-200
done; done; done
for res in 16; do for v in 4 16; do
  python scripts/bench_layer.py --res $res --cin 512 --cout 512 --batch 8 --prec 3 --ab 3 --va 0 --vb $v --rounds 5 --iters 50 2>/dev/null | cut -c1-200
done; done
for res in 8 4; do for v in 8; do
  python scripts/bench_layer.py --res $res --cin 512 --cout 512 --batch 8 --prec 3 --ab 3 --va 0 --vb $v --rounds 5 --iters 50 2>/dev/null | cut -c1-200
done; done
python scripts/bench_layer.py --res 64 --cin 512 --cout 512 --batch 8 --prec 3 --up --bwd --ab 3 --va 0 --vb 3 --rounds 5 --iters 50 2>/dev/null | cut -c1-200
python scripts/bench_layer.py --res 64 --cin 512 --cout 512 --batch 8 --prec 3 --up --ab 3 --va 0 --vb 2 --rounds 5 --iters 50 2>/dev/null | cut -c1-200
