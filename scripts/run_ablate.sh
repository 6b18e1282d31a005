cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in 0 1 2 3 4 5; do
  if [ $n = 0 ]; then unset LATENTAUG_HIP_LIB; else export LATENTAUG_HIP_LIB=build_variants/lib_ab$n.so; fi
  timeout -k 10 200 rocprofv3 --kernel-trace -d gpurun_out/ab$n -o r -- python3 scripts/bench_layer.py --prec 3 --iters 5 > /dev/null 2>&1 && echo "ablate $n" && python scripts/prof_summary.py gpurun_out/ab$n/r_results.db 1
done
