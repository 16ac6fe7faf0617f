#!/usr/bin/env bash
# LDS bank-conflict share and MFMA busy share per kernel of the bench workload: one rocprofv3 PMC pass per
# counter group (counters only, no trace domain beside --kernel-trace).  Usage: bash tools/pmc_lds.sh <tag>
set -u
TAG=${1:-pmc_lds}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT"; do
  d=$(echo $c | tr ' ' '_')
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$d" -- \
     python3 "$ROOT/tools/pmc_forward.py" > "$OUT/$d.log" 2>&1
  rc=$?
  echo "pass $c rc=$rc"
  if [ $rc -ne 0 ]; then tail -n 5 "$OUT/$d.log"; fi
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping"; exit $rc; fi
done
cd "$ROOT"
python3 - "$OUT" <<'PYEOF'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if n.startswith("afx::"):
            agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/pmc_lds.txt", "w") as fh:
    for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_LDS_IDX_ACTIVE", [0]))):
        m = {c: sum(v) / len(v) for c, v in d.items()}
        line = f"{k[:78]:78s} " + "  ".join(f"{c}={v:.3g}" for c, v in sorted(m.items()))
        if m.get("SQ_LDS_IDX_ACTIVE"):
            line += f"  conflict/active={m.get('SQ_LDS_BANK_CONFLICT', 0) / m['SQ_LDS_IDX_ACTIVE']:.3f}"
        if m.get("SQ_BUSY_CYCLES"):
            line += f"  mfma_busy/busy={m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / m['SQ_BUSY_CYCLES']:.3f}"
        print(line)
        fh.write(line + "\n")
PYEOF
find "$OUT" -name "*kernel_trace.csv" -delete 2>/dev/null
