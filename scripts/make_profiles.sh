# dev tool (GPU box): kernel-trace stats + the two PMC passes of the default bench -> gpurun_out/profiles_new/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=${1:-f16x2}
O=gpurun_out/profiles_new
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --precision $P > $O/bench_kt.log 2>&1 || exit 1
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$P.csv
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pf -o f -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --precision $P > $O/pf.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pw -o w -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --precision $P > $O/pw.log 2>&1 || exit 1
python scripts/pmc_traffic.py $O/pf/f_results.db $O/pw/w_results.db $P $O/pmc_traffic_$P.json
rm -rf $O/kt $O/pf $O/pw
