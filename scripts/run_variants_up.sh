# dev tool: flat-kernel time of the up-layer forward / backward shapes for each library variant given as argument
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in "$@"; do
  if [ "$L" = default ]; then unset LATENTAUG_HIP_LIB; else export LATENTAUG_HIP_LIB=build_variants/$L; fi
  i=0
  for args in "--res 256 --cin 256 --cout 128 --up" "--res 128 --cin 512 --cout 256 --up" "--res 256 --cin 256 --cout 128 --up --bwd"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace -d gpurun_out/varu$i -o r -- python3 scripts/bench_layer.py --prec 3 --iters 5 $args > /dev/null 2>&1 && echo "$L $args: $(python scripts/prof_summary.py gpurun_out/varu$i/r_results.db 4 | grep "la_conv_bf16_kernel" | head -1 | cut -c1-75)"
    rm -rf gpurun_out/varu$i
  done
done
