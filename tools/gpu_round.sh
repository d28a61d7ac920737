#!/bin/bash
# One GPU-box call: parity tests, bench line, rocprofv3 kernel stats, PMC passes (HBM bytes, SQ).
# Usage (via gpurun): bash tools/gpu_round.sh <tag> [tests|notests]
# Everything lands under gpurun_out/<tag>/; copy what should be judged into profiles/<tag>/.
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?}"
TAG=${1:-run}
TESTS=${2:-tests}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
if [ "$TESTS" = tests ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > "$OUT/pytest.log" 2>&1 || { tail -30 "$OUT/pytest.log"; exit 1; }
  tail -3 "$OUT/pytest.log"
fi
timeout -k 10 300 python bench.py --steps 20 --warmup 3 > "$OUT/bench_line.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }
cat "$OUT/bench_line.json"
# per-kernel time
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -o s -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-leg --no-legs --no-inflight-leg > "$OUT/prof_stats.log" 2>&1
# HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots), then SQ counters
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-leg --no-legs --no-inflight-leg > "$OUT/pmc_fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$OUT/pmc_write" -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-leg --no-legs --no-inflight-leg > "$OUT/pmc_write.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d "$OUT/pmc_sq" -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-leg --no-legs --no-inflight-leg > "$OUT/pmc_sq.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/pmc_sq2" -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-leg --no-legs --no-inflight-leg > "$OUT/pmc_sq2.log" 2>&1 || true
find "$OUT/prof_stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
# ---- BASELINE configs[2] with the errors left in (k = 51: the two-word keys, the k-mer-level repartition): kernel times and
# HBM traffic of k_ovf_scatter / k_count_buckets<2>
K51="--k 51 --err 0.01 --steps 3 --warmup 1 --no-cpu-baseline --no-host-leg --no-legs --no-inflight-leg"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/k51_stats" -o s -- python bench.py $K51 > "$OUT/k51_stats.log" 2>&1 || true
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$OUT/k51_pmc_fetch" -o p -- python bench.py $K51 > "$OUT/k51_pmc_fetch.log" 2>&1 || true
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$OUT/k51_pmc_write" -o p -- python bench.py $K51 > "$OUT/k51_pmc_write.log" 2>&1 || true
find "$OUT/k51_stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_config2_errors_left_in_k51.csv" \;
python tools/pmc_summary.py "$OUT/k51_pmc_*/**/*counter_collection.csv" > "$OUT/pmc_summary_config2_errors_left_in_k51.txt" 2>&1 || true
# ---- GPU idle time between the kernels of a step: one handle at a time, and with two handles in flight
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace1" -o t -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --no-legs --no-inflight-leg > "$OUT/trace1.log" 2>&1 || true
find "$OUT/trace1" -name "*kernel_trace.csv" -exec python tools/gap_report.py {} \; > "$OUT/gpu_idle_gaps.txt" 2>&1 || true
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace2" -o t -- python tools/two_in_flight.py 12 2 > "$OUT/two_in_flight.log" 2>&1 || true
find "$OUT/trace2" -name "*kernel_trace.csv" -exec python tools/busy_report.py {} 8 \; > "$OUT/gpu_busy_two_in_flight.txt" 2>&1 || true
cat "$OUT/two_in_flight.log" >> "$OUT/gpu_busy_two_in_flight.txt" || true
python tools/pmc_summary.py "$OUT/pmc_*/**/*counter_collection.csv" > "$OUT/pmc_summary.txt" 2>&1 || true
# keep the merge-back small: drop the raw traces
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
find "$OUT" -name "*.db" -delete; find "$OUT" -name "*counter_collection.csv" -delete
du -sh "$OUT"
