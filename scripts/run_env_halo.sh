# dev tool: halo-kernel time of four 154.6-GFLOP launches under each environment setting given as argument ("none" = default)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for E in "$@"; do
  [ "$E" != none ] && export $E
  i=0
  for args in "--res 256 --cin 128 --cout 128" "--res 128 --cin 256 --cout 256" "--res 64 --cin 512 --cout 512" "--res 256 --cin 128 --cout 128 --bwd" "--res 128 --cin 256 --cout 256 --bwd" "--res 64 --cin 512 --cout 512 --bwd"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace -d gpurun_out/varh$i -o r -- python3 scripts/bench_layer.py --prec 3 --iters 5 $args > /dev/null 2>&1 && echo "$E $args: $(python scripts/prof_summary.py gpurun_out/varh$i/r_results.db 4 | grep "la_conv_bf16_halo" | head -1 | cut -c1-75)"
    rm -rf gpurun_out/varh$i
  done
  [ "$E" != none ] && unset ${E%%=*}
done
