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
# what the two host-level schedules of this round are worth on this box: one loop instead of two stream lanes, whole frames in every loop step
: > $O/${RR}_bench_variants.jsonl
for V in "--lanes 1" "--whole-frames" "--lanes 1 --whole-frames"; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline $V 2>/dev/null | tail -1 >> $O/${RR}_bench_variants.jsonl
done
bash scripts/run_other_shapes.sh; cp gpurun_out/other_shapes.jsonl $O/${RR}_bench_other_shapes.jsonl
# per-shape kernel rows of the other workloads (eager launches, criteria one after the other, lanes one after the other)
bash scripts/run_trace_cfg.sh wdisc --w-disc 0.01 && cp gpurun_out/shapes_wdisc.csv $O/${RR}_kernel_shapes_wdisc.csv
bash scripts/run_trace_cfg.sh E --preset E --no-overlap && cp gpurun_out/shapes_E.csv $O/${RR}_kernel_shapes_E.csv
bash scripts/run_trace_cfg.sh c512 --res 512 --batch 4 && cp gpurun_out/shapes_c512.csv $O/${RR}_kernel_shapes_c512.csv
bash scripts/run_trace_cfg.sh d1024 --res 1024 --batch 2 --latent-steps 50 && cp gpurun_out/shapes_d1024.csv $O/${RR}_kernel_shapes_d1024.csv
python bench.py --gpus 2 --dist-backend gloo --force-device 0 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/${RR}_bench_selflaunch_2ranks_1gpu_gloo.json 2> $O/bench_g2.err
ls -la $O
