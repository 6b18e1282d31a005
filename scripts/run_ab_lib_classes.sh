set -e
pr() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', round(d['ms_per_step'],2), {k:round(v['ms_per_batch'],2) for k,v in r['classes'].items()})"; }
python scripts/bench_with_lib.py tmp_libs/base/liblatentaug_hip.so --lanes 1 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | pr base
python bench.py --lanes 1 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | pr new
