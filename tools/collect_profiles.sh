#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash tools/collect_profiles.sh <tag>'): the bench line, a rocprofv3 kernel-trace/--stats run of the same
# command and three PMC passes (MFMA counters, FETCH_SIZE, WRITE_SIZE -- separate passes, never combined with sys/runtime traces), all
# under gpurun_out/<tag>/.  tools/prof_summary.py and tools/pmc_summary.py turn them into the files committed under profiles/.
# A failed or timed-out step ends the script: no GPU step is started after one that did not finish.
set -e -o pipefail
TAG=${1:-r2p}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
echo "[1/5] bench line (with CPU baseline)"
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_line.json" 2> "$OUT/bench.err"
cut -c1-160 "$OUT/bench_line.json"
echo "[2/5] kernel trace + stats of the bench command"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > "$OUT/stats_line.json" 2> "$OUT/stats.err"
PMC="python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline"
echo "[3/5] PMC: MFMA"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/mfma" -o run -- $PMC > "$OUT/pmc_mfma.log" 2>&1
echo "[4/5] PMC: FETCH_SIZE"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o run -- $PMC > "$OUT/pmc_fetch.log" 2>&1
echo "[5/5] PMC: WRITE_SIZE"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o run -- $PMC > "$OUT/pmc_write.log" 2>&1
# only the per-kernel tables travel back (the traces themselves are large)
find "$OUT" -name '*_kernel_trace.csv' -size +20M -delete || true
ls -la "$OUT" "$OUT"/stats "$OUT"/mfma
