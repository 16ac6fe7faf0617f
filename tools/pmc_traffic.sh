#!/usr/bin/env bash
# HBM traffic per kernel launch of the bench workload: two rocprofv3 PMC passes (FETCH_SIZE,
# WRITE_SIZE -- they do not fit one pass; counters only, no trace domain beside --kernel-trace),
# corrected as MI355X_MICROARCH.md "HBM" prescribes (gfx950: FETCH_SIZE x 2; both in KB).
# Usage: [AFX_WORKLOAD=xlsr_aasist] bash tools/pmc_traffic.sh <tag>   ->  gpurun_out/<tag>/pmc_traffic.json
set -u
TAG=${1:-pmc_traffic}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$c" -- \
     python3 "$ROOT/tools/pmc_forward.py" > "$OUT/$c.log" 2>&1
  rc=$?
  echo "pass $c rc=$rc"
  if [ $rc -ne 0 ]; then tail -n 5 "$OUT/$c.log"; fi
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping"; exit $rc; fi
done
cd "$ROOT"
python3 - "$OUT" <<'PYEOF'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not n.startswith("afx::"):
            continue
        agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in sorted(agg.items()):
    f, w = d.get("FETCH_SIZE", []), d.get("WRITE_SIZE", [])
    if not f or not w:
        continue
    res[k] = {"launches_sampled": len(f), "fetch_bytes_per_launch": 2 * 1024 * sum(f) / len(f),
              "write_bytes_per_launch": 1024 * sum(w) / len(w)}
    print(f"{k[:70]:70s} n={len(f):4d} fetch {res[k]['fetch_bytes_per_launch']/1e6:9.1f} MB  write {res[k]['write_bytes_per_launch']/1e6:9.1f} MB")
import os
stamp = {}
try:
    stamp = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(out))), "real-time-deepfake-speech-detection_amd", "lib", "build_stamp.json")))
except Exception:
    pass
json.dump({"git": stamp.get("git", "unknown"), "git_dirty": stamp.get("dirty"), "build_id": stamp.get("build_id", "unknown"), "dtype": os.environ.get("AFX_DTYPE", "fp16"), "note": "FETCH_SIZE x2 (gfx950 correction) and WRITE_SIZE, KB -> bytes, mean per launch over 3 forwards of "
                   "workload " + os.environ.get("AFX_WORKLOAD", "conformer_student") + " (tools/pmc_forward.py)", "kernels": res}, open(out + "/pmc_traffic.json", "w"), indent=1)
PYEOF
find "$OUT" -name "*counter_collection.csv" -size +20M -delete 2>/dev/null
exit 0
