#!/usr/bin/env bash
# rocprofv3 PMC passes over tools/pmc_gemm.py (counters only: no trace domains beside
# --kernel-trace, one counter group per pass).  Usage: bash tools/pmc.sh <tag>
set -u
TAG=${1:-pmc}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/p$i" -- \
     python3 "$ROOT/tools/pmc_gemm.py" > "$OUT/p$i.log" 2>&1
  rc=$?
  echo "pass $i ($grp) rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping"; exit $rc; fi
done
cd "$ROOT"
python3 - "$OUT" <<'EOF'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gemm_kernel" not in n:
            continue
        key = n.split("gemm_kernel<")[1].split(">")[0]
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print("==", k)
    for c, v in sorted(d.items()):
        print(f"   {c:34s} mean {sum(v)/len(v):.4g}  (n={len(v)})")
EOF
