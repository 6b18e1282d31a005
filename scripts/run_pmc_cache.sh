# dev tool: L1->L2 request and L2 hit counters of the contraction kernels over one batch of the default bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d gpurun_out/pc -o c -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-graph --lanes-serial > gpurun_out/pc.log 2>&1 || { tail -5 gpurun_out/pc.log; exit 1; }
python scripts/pmc_generic.py gpurun_out/pc/c_results.db la_conv > gpurun_out/pmc_cache.txt
rm -rf gpurun_out/pc
