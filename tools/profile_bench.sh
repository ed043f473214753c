#!/bin/bash
# Profiles bench.py on the GPU box and leaves only text summaries under gpurun_out/$1:
#   bench.json (plain run), kernel_stats.csv (rocprofv3 --kernel-trace of the same command),
#   pmc_hbm.csv (FETCH_SIZE / WRITE_SIZE, one pass each), pmc_sq.csv (SQ counters).
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p "$OUT"
T=$GRAFT_REPO_ROOT/tools/rocpd_summary.py
timeout -k 10 300 python3 $GRAFT_REPO_ROOT/bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pb_* 
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/pb_kt -o b -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> /dev/null || exit 1
python3 $T kernels $(find /tmp/pb_kt -name "*.db" | head -1) > "$OUT/kernel_stats.csv" || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d /tmp/pb_f -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline > /dev/null 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d /tmp/pb_w -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline > /dev/null 2>&1 || exit 1
(python3 $T counters $(find /tmp/pb_f -name "*.db" | head -1) | grep "Kernel\|ccnn::"; python3 $T counters $(find /tmp/pb_w -name "*.db" | head -1) | grep "ccnn::") > "$OUT/pmc_hbm.csv"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pb_s -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline > /dev/null 2>&1 || exit 1
python3 $T counters $(find /tmp/pb_s -name "*.db" | head -1) | grep "Kernel\|ccnn::" > "$OUT/pmc_sq.csv"
ls -la "$OUT"
