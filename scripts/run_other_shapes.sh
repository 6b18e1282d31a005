# dev tool (GPU box): bench lines of the non-default workloads -> gpurun_out/other_shapes.jsonl
cd $GRAFT_REPO_ROOT
O=gpurun_out/other_shapes.jsonl
: > $O
python bench.py --res 512 --batch 4 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 >> $O
python bench.py --res 1024 --batch 2 --latent-steps 50 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 >> $O
python bench.py --batch 2 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 >> $O
python bench.py --batch 4 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 >> $O
python bench.py --preset E --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 >> $O
python bench.py --w-disc 0.01 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 >> $O
