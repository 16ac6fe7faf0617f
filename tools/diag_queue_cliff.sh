#!/usr/bin/env bash
# the two-stream teacher step next to RCCL ran 2x SLOWER than the one-stream step with
# GPU_MAX_HW_QUEUES=8 (or a high-priority side stream) -- profiles/r03_k_dist_overlap_hw_queues.txt recorded the correlation
# ("more than four hardware queues in use"), not the cause.  This takes one rocprofv3 --kernel-trace of the step in the healthy
# setting and one in the slow setting (program directly behind `--`, the two-stream form forced) and prints, per kernel and per
# hardware queue, launch counts and average durations, plus how much of the wall time each queue's kernels cover.
# Usage: bash tools/diag_queue_cliff.sh <tag>   ->  gpurun_out/<tag>/queue_cliff.txt
set -u
TAG=${1:-queue_cliff}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export AFX_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1
# first WITHOUT the profiler: does the cliff exist on this box / build at all?  (two-stream forced, then one-stream, per setting)
for mode in healthy slow; do
  if [ $mode = slow ]; then export GPU_MAX_HW_QUEUES=8; else unset GPU_MAX_HW_QUEUES; fi
  for form in --force-overlap --no-overlap; do
    export MASTER_PORT=$((29700 + RANDOM % 200))
    timeout -k 10 200 python3 "$ROOT/bench.py" --workload xlsr_aasist --no-config3 --cpu-sample 0 --steps 10 --warmup 3 $form 2> /dev/null | grep '^{' > "$OUT/plain_${mode}${form}.json"
    echo "no profiler, $mode (GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-default}), $form: $(python3 -c "import json; d=json.load(open('$OUT/plain_${mode}${form}.json')); print(d['value'], 'utt/s', d['ms_per_step'], 'ms per step')" 2>/dev/null)" | tee -a "$OUT/queue_cliff_plain.txt"
  done
done
for mode in healthy slow; do
  export MASTER_PORT=$((29700 + RANDOM % 200))
  if [ $mode = slow ]; then export GPU_MAX_HW_QUEUES=8; else unset GPU_MAX_HW_QUEUES; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/$mode" -- \
    python3 "$ROOT/bench.py" --workload xlsr_aasist --no-config3 --cpu-sample 0 --steps 10 --warmup 3 --force-overlap > "$OUT/$mode.json" 2> "$OUT/$mode.err"
  rc=$?
  echo "under rocprofv3, $mode rc=$rc: $(grep '^{' "$OUT/$mode.json" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'utt/s', d['ms_per_step'], 'ms per step')" 2>/dev/null)" | tee -a "$OUT/queue_cliff_plain.txt"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping"; exit $rc; fi
done
cd "$ROOT"
python3 - "$OUT" <<'PYEOF' | tee "$OUT/queue_cliff.txt"
import collections, csv, glob, sys
out = sys.argv[1]
for mode in ("healthy", "slow"):
    rows = []
    for f in glob.glob(f"{out}/{mode}/**/*kernel_trace.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    if not rows:
        print(mode, "no trace")
        continue
    cols = rows[0].keys()
    qk = "Queue_Id" if "Queue_Id" in cols else None
    # the timed region = the last 10 steps' worth of launches: take the last 60 % of the trace by time
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    t1 = max(int(r["End_Timestamp"]) for r in rows)
    cut = t1 - int(0.35 * (t1 - t0))
    rows = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
    span = (max(int(r["End_Timestamp"]) for r in rows) - min(int(r["Start_Timestamp"]) for r in rows)) / 1e6
    per = collections.defaultdict(list)
    perq = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
        q = r[qk] if qk else "?"
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        per[(n, q)].append(d)
        perq[q][0] += 1
        perq[q][1] += d / 1e3
    print(f"== {mode}: {len(rows)} launches in the last {span:.1f} ms of the trace; per queue: " + ", ".join(f"queue {q}: {c} launches, {ms:.1f} ms of kernel time" for q, (c, ms) in sorted(perq.items())))
    for (n, q), v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:14]:
        print(f"   queue {q:>3s} {n:60s} n={len(v):4d} avg {sum(v)/len(v):8.1f} us  total {sum(v)/1e3:7.2f} ms")
PYEOF
find "$OUT" -name "*kernel_trace.csv" -delete 2>/dev/null
exit 0
