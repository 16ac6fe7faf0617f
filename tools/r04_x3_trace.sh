#!/usr/bin/env bash
# Kernel trace of one bench leg in a chosen precision, per (kernel, grid) averages.
# usage: bash tools/r04_x3_trace.sh <tag> <workload> <dtype>
set -u
TAG=$1; WL=$2; DT=$3
ROOT=$(pwd); OUT=gpurun_out/$TAG; mkdir -p "$OUT"
cp real-time-deepfake-speech-detection_amd/lib/build_stamp.json "$OUT/build_stamp.json" 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/prof" -- \
  python3 "$ROOT/bench.py" --workload $WL --dtype $DT --steps 10 --warmup 3 --cpu-sample 0 --no-config3 > "$ROOT/$OUT/prof.log" 2>&1
rc=$?
cd "$ROOT"
tail -n 2 "$OUT/prof.log"
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
python3 - "$OUT" <<'PYEOF'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/prof/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[(n, r["Grid_Size_X"], r["Grid_Size_Z"], r["Workgroup_Size_X"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
stamp = open(out + "/build_stamp.json").read().strip()
tot = sum(sum(v) for v in agg.values())
with open(out + "/kernel_by_shape.csv", "w") as fh:
    fh.write("# build " + stamp + "\n")
    fh.write("kernel,grid_x_threads,grid_z,wg_size,calls,avg_us,total_ms,share\n")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        fh.write(f'"{k[0]}",{k[1]},{k[2]},{k[3]},{len(v)},{sum(v)/len(v)/1e3:.1f},{sum(v)/1e6:.3f},{sum(v)/tot:.4f}\n')
PYEOF
find "$OUT/prof" -name "*kernel_trace.csv" -delete 2>/dev/null
exit 0
