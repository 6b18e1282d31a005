# dev tool: contraction-kernel time of six probe layers under each environment setting given as argument ("none" = default)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for E in "$@"; do
  [ "$E" != none ] && export $E
  i=0
  for args in "--res 256 --cin 256 --cout 128 --up" "--res 128 --cin 512 --cout 256 --up" "--res 256 --cin 256 --cout 128 --up --bwd" "--res 128 --cin 512 --cout 256 --up --bwd" "--res 32 --cin 512 --cout 512" "--res 32 --cin 512 --cout 512 --bwd" "--res 16 --cin 512 --cout 512" "--res 32 --cin 512 --cout 512 --up" "--res 32 --cin 512 --cout 512 --up --bwd"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace -d gpurun_out/vare$i -o r -- python3 scripts/bench_layer.py --prec 3 --iters 5 $args > /dev/null 2>&1 && echo "$E $args: $(python scripts/prof_summary.py gpurun_out/vare$i/r_results.db 6 | grep "la_conv_bf16\|la_conv_lin" | head -1 | cut -c1-80)"
    rm -rf gpurun_out/vare$i
  done
  [ "$E" != none ] && unset ${E%%=*}
done
