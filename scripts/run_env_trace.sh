# dev tool: per-shape kernel rows of the default bench (eager launches) under each environment setting given as argument ("none" = default)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for E in "$@"; do
  i=$((i+1))
  [ "$E" != none ] && export $E
  timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/ab$i -o b -- python3 bench.py --dev-build --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-graph --lanes 1 > gpurun_out/ab$i.log 2>&1 || exit 1
  python scripts/shape_stats.py gpurun_out/ab$i/b_results.db 3 > gpurun_out/ab_shapes_$i.csv
  rm -rf gpurun_out/ab$i
  [ "$E" != none ] && unset ${E%%=*}
done
