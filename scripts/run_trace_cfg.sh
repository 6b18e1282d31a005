# dev tool: per-shape kernel rows of a bench configuration (eager launches).  usage: run_trace_cfg.sh TAG <bench args...>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=$1; shift
timeout -k 10 400 rocprofv3 --kernel-trace -d gpurun_out/tc_$T -o b -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-graph --lanes-serial "$@" > gpurun_out/tc_$T.log 2>&1 || exit 1
python scripts/shape_stats.py gpurun_out/tc_$T/b_results.db 2 > gpurun_out/shapes_$T.csv
rm -rf gpurun_out/tc_$T
