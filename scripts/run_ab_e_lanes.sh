# preset E: one loop with the in-graph criteria fork (default) against two lanes with the split replay (overlap mode 1), alternating, same box
set -e
for i in 1 2 3; do
for m in "--lanes 1" "--lanes 2 --overlap-mode 1"; do
python bench.py --preset E $m --no-cpu-baseline --no-roofline --steps 4 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', round(d['value'],2), round(d['ms_per_step'],1))"
done; done
