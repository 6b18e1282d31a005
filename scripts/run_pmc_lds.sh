# dev tool: LDS activity and bank conflicts of the contraction kernels, one stride-1 layer (forward and backward) and one up layer
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > gpurun_out/pmc_lds.txt
for R in "--res ${1:-128} --cin ${2:-256} --cout ${3:-256}" "--res ${1:-128} --cin ${2:-256} --cout ${3:-256} --bwd" "--res 128 --cin 256 --cout 256 --up" "--res 128 --cin 256 --cout 256 --up --bwd"; do
  echo "## $R" >> gpurun_out/pmc_lds.txt
  if timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d gpurun_out/pl -o c -- python3 scripts/bench_layer.py $R --batch 8 --prec 3 --product > gpurun_out/pl.log 2>&1; then
    python scripts/pmc_generic.py gpurun_out/pl/c_results.db la_conv_bf16 >> gpurun_out/pmc_lds.txt
  else echo failed >> gpurun_out/pmc_lds.txt; tail -3 gpurun_out/pl.log >> gpurun_out/pmc_lds.txt; fi
  rm -rf gpurun_out/pl
done
