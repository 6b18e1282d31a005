#!/bin/bash
# GPU box: every artefact of profiles/<RR>_* from ONE box at ONE commit -> gpurun_out/profiles_new/
#   usage: bash scripts/make_round_profiles.sh r04
RR=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/profiles_new
rm -rf $O; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/${RR}_bench_default.json 2> $O/bench_default.err || exit 1
bash scripts/make_profiles.sh f16x2 $RR > $O/make_profiles.log 2>&1 || { tail -5 $O/make_profiles.log; exit 1; }
bash scripts/run_pmc_cache.sh > $O/pmc_cache.log 2>&1 && cp gpurun_out/pmc_cache.txt $O/${RR}_pmc_cache_f16x2.txt
# the bench line again, now that profiles/<RR>_pmc_traffic exists next to it (roofline.traffic is read from that file)
cp $O/${RR}_pmc_traffic_f16x2.json profiles/ 2>/dev/null
python bench.py --steps 20 --warmup 5 > $O/${RR}_bench_default.json 2> $O/bench_default.err || exit 1
# exact-fp32 contraction, same workload (what the f16x2 emulation is a speed-up of)
python bench.py --steps 4 --warmup 1 --precision f32 --no-cpu-baseline > $O/${RR}_bench_precision_f32.json 2> $O/bench_f32.err
bash scripts/run_other_shapes.sh; cp gpurun_out/other_shapes.jsonl $O/${RR}_bench_other_shapes.jsonl
python bench.py --gpus 2 --dist-backend gloo --force-device 0 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/${RR}_bench_selflaunch_2ranks_1gpu_gloo.json 2> $O/bench_g2.err
ls -la $O
