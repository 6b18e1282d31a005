#!/bin/bash
# experiment: do two (four) independent processes with half (quarter) the batch each overlap well enough on one GPU to beat one process with
# the whole batch?  (the latency-bound low-resolution trunk of one would run under the big kernels of the other)
mkdir -p gpurun_out/r3d
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r3d/b8.json 2>/dev/null
for n in 2 4; do
  b=$((8 / n))
  pids=""
  for i in $(seq 1 $n); do
    python bench.py --batch $b --steps 30 --warmup 8 --no-cpu-baseline --no-roofline > gpurun_out/r3d/b${b}_$i.json 2>/dev/null &
    pids="$pids $!"
  done
  wait $pids
done
python - <<'PY'
import json, glob
def v(f):
    try: return json.loads(open(f).read().strip().splitlines()[-1])['value']
    except Exception as e: return float('nan')
print('one process B=8:', v('gpurun_out/r3d/b8.json'))
for b in (4, 2):
    vs = [v(f) for f in sorted(glob.glob(f'gpurun_out/r3d/b{b}_*.json'))]
    print(f'{len(vs)} concurrent processes B={b}: each', vs, 'sum', sum(vs))
PY
