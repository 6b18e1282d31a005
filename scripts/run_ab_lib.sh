# same-box A/B of the shipped library against tmp_libs/base/liblatentaug_hip.so (a copy of another build; *.so is git-ignored but travels to the GPU box)
set -e
for i in 1 2; do
python scripts/bench_with_lib.py tmp_libs/base/liblatentaug_hip.so --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('base lanes', round(d['ms_per_step'],2))"
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new  lanes', round(d['ms_per_step'],2), d['witness']['ok'])"
python scripts/bench_with_lib.py tmp_libs/base/liblatentaug_hip.so --lanes 1 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('base one', round(d['ms_per_step'],2))"
python bench.py --lanes 1 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new  one', round(d['ms_per_step'],2))"
done
