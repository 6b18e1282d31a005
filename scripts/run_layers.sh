# dev tool: kernel-level timing of three representative layers (rocprofv3 kernel trace), optional LATENTAUG_HIP_LIB
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=${1:-3}
i=0
for args in "--res 256 --cin 128 --cout 128" "--res 64 --cin 512 --cout 512" "--res 128 --cin 256 --cout 256 --bwd"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace -d gpurun_out/lay$i -o r -- python3 scripts/bench_layer.py --prec $P --iters 5 $args > /dev/null 2>&1 && echo "layer $args" && python scripts/prof_summary.py gpurun_out/lay$i/r_results.db 3 | grep -v "total\|distribution"
done
