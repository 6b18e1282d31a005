# dev tool: where the waves of the 64-row halo kernel wait (SQ) and how busy the vector-memory path is (TA / TCP / TD), one layer
# usage: run_pmc_tcp.sh RES CIN COUT [extra bench_layer.py flags]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R="--res ${1:-512} --cin ${2:-64} --cout ${3:-64} --batch 8 --prec 3 ${@:4}"
: > gpurun_out/pmc_tcp.txt
pass() {
  n=$1; shift
  if timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace -d gpurun_out/$n -o c -- python3 scripts/bench_layer.py $R > gpurun_out/$n.log 2>&1; then
    python scripts/pmc_generic.py gpurun_out/$n/c_results.db halo >> gpurun_out/pmc_tcp.txt
  else echo "pass $n ($*) failed" >> gpurun_out/pmc_tcp.txt; fi
  rm -rf gpurun_out/$n
}
pass p1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
pass p2 TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE
pass p3 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
pass p4 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TD_TD_BUSY_sum
pass p5 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS
