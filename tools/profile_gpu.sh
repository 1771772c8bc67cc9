#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel stats, the three PMC passes and the in-kernel phase stamps of the
# default bench; raw output under gpurun_out/, the summaries are copied into profiles/ by the caller.
#   gpurun --timeout 1100 -- 'bash tools/profile_gpu.sh r02'
set -e
TAG=${1:-run}
WLS=${2:-"bilstm3x500 literal deepspeech"}
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
for WL in $WLS; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$WL" -- python3 "$ROOT/bench.py" --workload $WL --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > "$OUT/stats_$WL.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch_$WL" -- python3 "$ROOT/bench.py" --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$OUT/fetch_$WL.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write_$WL" -- python3 "$ROOT/bench.py" --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$OUT/write_$WL.log" 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/mfma_$WL" -- python3 "$ROOT/bench.py" --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$OUT/mfma_$WL.log" 2>&1
  cp "$OUT"/stats_$WL/*/*kernel_stats.csv "$OUT/${TAG}_${WL}_kernel_stats.csv"
  python3 "$ROOT/tools/pmc_summary.py" $WL "$OUT/fetch_$WL" "$OUT/write_$WL" "$OUT/pmc_traffic.json" "$OUT/mfma_$WL" > "$OUT/${TAG}_pmc_$WL.txt"
  # the raw counter tables are large: keep the summaries only
  rm -rf "$OUT/fetch_$WL" "$OUT/write_$WL" "$OUT/mfma_$WL" "$OUT"/stats_$WL/*/*kernel_trace.csv
  echo "done $WL"
done
cd "$ROOT"
if [ -x tools/sb_st ]; then
  timeout -k 5 60 tools/sb_st 500 16 500 2 0 > "$OUT/${TAG}_persist_stamps.log" 2>&1
  python3 tools/stamps_to_json.py bilstm3x500 "$OUT/${TAG}_persist_stamps.log" "$OUT/persist_stamps.json" > /dev/null
fi
if [ -x tools/sb_wide_st ]; then   # wide persistent kernels (Hp 2048): hipcc ... -DNASR_WSTAMP=1 tools/widebench.hip lstm.hip lstm_wide.hip
  WIDE_STAMPS=1 timeout -k 5 120 tools/sb_wide_st 500 32 0 > "$OUT/${TAG}_wide_stamps.log" 2>&1
  python3 tools/stamps_to_json.py --wide deepspeech "$OUT/${TAG}_wide_stamps.log" "$OUT/persist_stamps.json" > /dev/null
fi
head -8 "$OUT/${TAG}_bilstm3x500_kernel_stats.csv"
cat "$OUT/${TAG}_pmc_bilstm3x500.txt"
