#!/bin/bash
# Profiles bench.py on the GPU box and leaves only text summaries under gpurun_out/$1:
#   bench.json (plain run, all extras), kernel_stats.csv (rocprofv3 --kernel-trace --stats of the
#   headline run), pmc_hbm.csv (FETCH_SIZE / WRITE_SIZE, one pass each), pmc_sq.csv (SQ counters),
#   the same three for the batched path at B = 2048 (tools/run_steps.py): *_b2048.csv, and the HBM
#   counters alone for fp64 at B = 100 and fp32 at B = 512 (both the fused kernel): pmc_hbm_f64_b100.csv, pmc_hbm_f32_b512.csv.
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p "$OUT"
T=$GRAFT_REPO_ROOT/tools/rocpd_summary.py
B=$GRAFT_REPO_ROOT/bench.py
R=$GRAFT_REPO_ROOT/tools/run_steps.py
timeout -k 10 500 python3 $B > "$OUT/bench.json" 2> "$OUT/bench.err" || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pb_*
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/pb_kt -o b -- python3 $B --no-cpu-baseline --no-extras > "$OUT/bench_under_rocprof.json" 2> /dev/null || exit 1
python3 $T kernels $(find /tmp/pb_kt -name "*.db" | head -1) > "$OUT/kernel_stats.csv" || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d /tmp/pb_f -o p -- python3 $B --steps 200 --warmup 20 --no-cpu-baseline --no-extras > /dev/null 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d /tmp/pb_w -o p -- python3 $B --steps 200 --warmup 20 --no-cpu-baseline --no-extras > /dev/null 2>&1 || exit 1
(python3 $T counters $(find /tmp/pb_f -name "*.db" | head -1) | grep "Kernel\|ccnn::"; python3 $T counters $(find /tmp/pb_w -name "*.db" | head -1) | grep "ccnn::") > "$OUT/pmc_hbm.csv"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pb_s -o p -- python3 $B --steps 200 --warmup 20 --no-cpu-baseline --no-extras > /dev/null 2>&1 || exit 1
python3 $T counters $(find /tmp/pb_s -name "*.db" | head -1) | grep "Kernel\|ccnn::" > "$OUT/pmc_sq.csv"
# the batched path
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d /tmp/pb_kt2 -o b -- python3 $R pavia_u 2048 200 > /dev/null 2>&1 || exit 1
python3 $T kernels $(find /tmp/pb_kt2 -name "*.db" | head -1) > "$OUT/kernel_stats_b2048.csv" || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d /tmp/pb_f2 -o p -- python3 $R pavia_u 2048 60 > /dev/null 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d /tmp/pb_w2 -o p -- python3 $R pavia_u 2048 60 > /dev/null 2>&1 || exit 1
(python3 $T counters $(find /tmp/pb_f2 -name "*.db" | head -1) | grep "Kernel\|ccnn::"; python3 $T counters $(find /tmp/pb_w2 -name "*.db" | head -1) | grep "ccnn::") > "$OUT/pmc_hbm_b2048.csv"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pb_s2 -o p -- python3 $R pavia_u 2048 60 > /dev/null 2>&1 || exit 1
python3 $T counters $(find /tmp/pb_s2 -name "*.db" | head -1) | grep "Kernel\|ccnn::" > "$OUT/pmc_sq_b2048.csv"
# the canonical fp64 trainer's batch, and the largest batch of the fused kernel in fp32: HBM bytes only
for W in "100 f64" "512 f32"; do
    set -- $W
    timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d /tmp/pb_f_$1 -o p -- python3 $R pavia_u $1 100 $2 > /dev/null 2>&1 || exit 1
    timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d /tmp/pb_w_$1 -o p -- python3 $R pavia_u $1 100 $2 > /dev/null 2>&1 || exit 1
    (python3 $T counters $(find /tmp/pb_f_$1 -name "*.db" | head -1) | grep "Kernel\|ccnn::"; python3 $T counters $(find /tmp/pb_w_$1 -name "*.db" | head -1) | grep "ccnn::") > "$OUT/pmc_hbm_$2_b$1.csv"
done
ls -la "$OUT"
