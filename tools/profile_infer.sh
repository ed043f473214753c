#!/bin/bash
# Kernel trace + SQ counters of tools/bench_infer.py on the GPU box -> gpurun_out/$1 (text only).
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
shift
mkdir -p "$OUT"
T=$GRAFT_REPO_ROOT/tools/rocpd_summary.py
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pi_*
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/pi_kt -o b -- python3 $GRAFT_REPO_ROOT/tools/bench_infer.py "$@" > "$OUT/run.txt" 2> /dev/null || exit 1
python3 $T kernels $(find /tmp/pi_kt -name "*.db" | head -1) > "$OUT/kernel_stats.csv" || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pi_s -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_infer.py "$@" > /dev/null 2>&1 || exit 1
python3 $T counters $(find /tmp/pi_s -name "*.db" | head -1) | grep "Kernel\|ccnn::" > "$OUT/pmc_sq.csv"
cat "$OUT/run.txt"; cat "$OUT/kernel_stats.csv" | head -8; cat "$OUT/pmc_sq.csv"
