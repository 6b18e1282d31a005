# dev tool: per-shape kernel rows of a bench configuration under two environments.  usage: run_env_trace_cfg.sh ENV_A ENV_B <bench args...>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
EA=$1; EB=$2; shift; shift
i=0
for E in "$EA" "$EB"; do
  i=$((i+1))
  [ "$E" != none ] && export $E
  timeout -k 10 400 rocprofv3 --kernel-trace -d gpurun_out/ab$i -o b -- python3 bench.py --dev-build --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-graph --lanes 1 "$@" > gpurun_out/ab$i.log 2>&1 || exit 1
  python scripts/shape_stats.py gpurun_out/ab$i/b_results.db 2 > gpurun_out/ab_shapes_$i.csv
  rm -rf gpurun_out/ab$i
  [ "$E" != none ] && unset ${E%%=*}
done
