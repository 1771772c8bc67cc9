#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel stats and the two PMC passes of the default bench and the
# literal workload; raw output under gpurun_out/, the summaries are copied into profiles/ by the caller.
#   gpurun --timeout 900 -- 'bash tools/profile_gpu.sh r01_v5'
set -e
TAG=${1:-run}
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
for WL in bilstm3x500 literal; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$WL" -- python3 "$ROOT/bench.py" --workload $WL --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/stats_$WL.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch_$WL" -- python3 "$ROOT/bench.py" --workload $WL --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/fetch_$WL.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write_$WL" -- python3 "$ROOT/bench.py" --workload $WL --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/write_$WL.log" 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/mfma_$WL" -- python3 "$ROOT/bench.py" --workload $WL --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/mfma_$WL.log" 2>&1
  cp "$OUT"/stats_$WL/*/*kernel_stats.csv "$OUT/${TAG}_${WL}_kernel_stats.csv"
  python3 "$ROOT/tools/pmc_summary.py" $WL "$OUT/fetch_$WL" "$OUT/write_$WL" "$OUT/pmc_traffic.json" "$OUT/mfma_$WL" > "$OUT/pmc_$WL.txt"
  # the raw counter tables are large: keep the summaries only
  rm -rf "$OUT/fetch_$WL" "$OUT/write_$WL" "$OUT/mfma_$WL" "$OUT"/stats_$WL/*/*kernel_trace.csv
done
head -8 "$OUT/${TAG}_bilstm3x500_kernel_stats.csv"
cat "$OUT/pmc_bilstm3x500.txt"
