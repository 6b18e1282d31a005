# dev tool: kernel-trace durations of the small-grid layers under two settings of LA_DEV_KNOBS:  run_trace_small.sh "3=8" ""
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
KA=${1:-3=8}; KB=${2:-}
i=0
for args in "--res 4" "--res 8" "--res 16" "--res 16 --bwd" "--res 32" "--res 32 --bwd" "--res 16 --up" "--res 32 --up" "--res 64 --up --bwd"; do
  for K in "$KA" "$KB"; do
    i=$((i+1))
    LA_DEV_KNOBS="$K" timeout -k 10 200 rocprofv3 --kernel-trace -d gpurun_out/ts$i -o r -- python3 scripts/bench_layer.py --prec 3 --cin 512 --cout 512 --batch 8 --iters 20 $args > /dev/null 2>&1 && echo "layer $args knobs [$K]" && python scripts/prof_summary.py gpurun_out/ts$i/r_results.db 5 | grep "la_" | grep -v "pack\|absmax"; rm -rf gpurun_out/ts$i
  done
done
