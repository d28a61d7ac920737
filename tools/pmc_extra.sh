#!/bin/bash
# extra PMC passes for the two count kernels (LDS pressure, scalar cycles, VALU mix)
TAG=${1:-pmc_extra}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN -d "$OUT/p1" -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/p1.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INST_CYCLES_SALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_VALU2 SQ_LEVEL_WAVES -d "$OUT/p2" -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/p2.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc VALUBusy -d "$OUT/p3" -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/p3.log" 2>&1 || true
python tools/pmc_summary.py "$OUT/p*/**/*counter_collection.csv" > "$OUT/pmc_extra.txt" 2>&1 || true
find "$OUT" -name "*.csv" -delete; find "$OUT" -name "*.db" -delete
