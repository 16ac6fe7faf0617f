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
WL=${WORKLOAD:-conformer_student}   # bench.py --workload (xlsr_aasist = BASELINE config 3)
: > "$OUT/summary.txt"
# which build this visit measures (the GPU box has no .git: the Makefile recorded the commit and the source hash at build time)
echo "== build $(cat real-time-deepfake-speech-detection_amd/lib/build_stamp.json 2>/dev/null)" | tee -a "$OUT/summary.txt"
cp real-time-deepfake-speech-detection_amd/lib/build_stamp.json "$OUT/build_stamp.json" 2>/dev/null

if [ "${SKIP_TESTS:-0}" != "1" ]; then
echo "== pytest -m gpu $*" | tee -a "$OUT/summary.txt"
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 "$@" > "$OUT/pytest.log" 2>&1
rc=$?
tail -n 3 "$OUT/pytest.log" | tee -a "$OUT/summary.txt"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest killed (rc=$rc): stopping" | tee -a "$OUT/summary.txt"; exit $rc; fi
fi

echo "== bench.py (N=1, $WL)" | tee -a "$OUT/summary.txt"
timeout -k 10 600 python bench.py --workload $WL --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
rc=$?
cat "$OUT/bench.json" | tee -a "$OUT/summary.txt"
if [ $rc -ne 0 ]; then tail -n 20 "$OUT/bench.err" | tee -a "$OUT/summary.txt"; fi
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "bench killed (rc=$rc): stopping" | tee -a "$OUT/summary.txt"; exit $rc; fi

echo "== rocprofv3 --kernel-trace --stats" | tee -a "$OUT/summary.txt"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/prof" -- \
  python3 "$ROOT/bench.py" --workload $WL --steps 10 --warmup 3 --cpu-sample 0 > "$ROOT/$OUT/prof.log" 2>&1
rc=$?
cd "$ROOT"
STATS=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -n 1)
if [ -n "$STATS" ]; then head -n 25 "$STATS" | tee -a "$OUT/summary.txt"; else tail -n 20 "$OUT/prof.log" | tee -a "$OUT/summary.txt"; fi
# per-(kernel, grid) averages from the per-dispatch trace, then drop the (large) trace itself
python3 - "$OUT" <<'PYEOF'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
spans = []
for f in glob.glob(out + "/prof/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[(n, r["Grid_Size_X"], r["Grid_Size_Z"], r["Workgroup_Size_X"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        spans.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
# idle time between consecutive kernels of the stream (gaps under 100 us: back-to-back launches, not host pauses)
spans.sort()
gaps = [b[0] - a[1] for a, b in zip(spans, spans[1:]) if 0 <= b[0] - a[1] < 100000]
if gaps:
    gaps.sort()
    busy = sum(e - s for s, e in spans)
    with open(out + "/summary.txt", "a") as fh:
        fh.write(f"== inter-kernel gaps: {len(gaps)} gaps, median {gaps[len(gaps)//2]/1e3:.2f} us, mean {sum(gaps)/len(gaps)/1e3:.2f} us, "
                 f"sum {sum(gaps)/1e6:.2f} ms beside {busy/1e6:.2f} ms of kernel time ({sum(gaps)/(sum(gaps)+busy)*100:.1f} % idle)\n")
try:
    stamp = open(out + "/build_stamp.json").read().strip()
except OSError:
    stamp = "{}"
with open(out + "/kernel_by_shape.csv", "w") as fh:
    fh.write("# build " + stamp + "\n")
    fh.write("kernel,grid_x_threads,grid_z,wg_size,calls,avg_us,total_ms\n")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        fh.write(f'"{k[0]}",{k[1]},{k[2]},{k[3]},{len(v)},{sum(v)/len(v)/1e3:.1f},{sum(v)/1e6:.3f}\n')
PYEOF
find "$OUT/prof" -name "*kernel_trace.csv" -delete 2>/dev/null
exit 0
