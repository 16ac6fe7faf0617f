#!/usr/bin/env bash
# One GPU-box visit: parity tests, the bench line, and a rocprofv3 kernel-trace summary.
# Usage (from the repo root on the GPU box): bash tools/gpu_check.sh [tag] [pytest args...]
# A step that was killed or timed out ends the visit (no further GPU work after a hang).
set -u
TAG=${1:-run}
shift || true
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
ROOT=$(pwd)

echo "== pytest -m gpu $*" | tee "$OUT/summary.txt"
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 "$@" > "$OUT/pytest.log" 2>&1
rc=$?
tail -n 3 "$OUT/pytest.log" | tee -a "$OUT/summary.txt"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest killed (rc=$rc): stopping" | tee -a "$OUT/summary.txt"; exit $rc; fi

echo "== bench.py (N=1)" | tee -a "$OUT/summary.txt"
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
rc=$?
cat "$OUT/bench.json" | tee -a "$OUT/summary.txt"
if [ $rc -ne 0 ]; then tail -n 20 "$OUT/bench.err" | tee -a "$OUT/summary.txt"; fi
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "bench killed (rc=$rc): stopping" | tee -a "$OUT/summary.txt"; exit $rc; fi

echo "== rocprofv3 --kernel-trace --stats" | tee -a "$OUT/summary.txt"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/prof" -- \
  python3 "$ROOT/bench.py" --steps 10 --warmup 3 --cpu-sample 0 > "$ROOT/$OUT/prof.log" 2>&1
rc=$?
cd "$ROOT"
STATS=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -n 1)
if [ -n "$STATS" ]; then head -n 25 "$STATS" | tee -a "$OUT/summary.txt"; else tail -n 20 "$OUT/prof.log" | tee -a "$OUT/summary.txt"; fi
# keep the merge-back small: the per-dispatch trace can be large
find "$OUT/prof" -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
exit 0
