# dev tool (GPU box): artefacts under profiles/ for round RR of the default bench: rocprofv3 kernel-trace stats, per-shape rows,
# the two PMC traffic passes (per kernel class) and the MFMA-busy pass -> gpurun_out/profiles_new/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=${1:-f16x2}
RR=${2:-r02}
O=gpurun_out/profiles_new
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-graph --lanes-serial --precision $P > $O/bench_kt.log 2>&1 || exit 1
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/${RR}_kernel_stats_$P.csv
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/kd -o b -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-graph --lanes-serial --precision $P > $O/bench_kd.log 2>&1 || exit 1
python scripts/shape_stats.py $O/kd/b_results.db 4 > $O/${RR}_kernel_shapes_$P.csv
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pf -o f -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-graph --lanes-serial --precision $P > $O/pf.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pw -o w -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-graph --lanes-serial --precision $P > $O/pw.log 2>&1 || exit 1
python scripts/pmc_classes.py $O/pf/f_results.db $O/pw/w_results.db $P $O/${RR}_pmc_traffic_$P.json
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -d $O/pm -o m -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-graph --lanes-serial --precision $P > $O/pm.log 2>&1 || exit 1
python scripts/pmc_mfma.py $O/pm/m_results.db $O/${RR}_pmc_mfma_$P.json
rm -rf $O/kt $O/kd $O/pf $O/pw $O/pm
