#!/usr/bin/env bash
# Matrix-pipe utilisation per kernel of the bench workload as a 0..1 FRACTION, plus the LDS bank-conflict share:
# rocprofv3 PMC passes (counters only, no trace domain beside --kernel-trace; one pass per group).
#   mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x SIMDs of the chip)
#     SQ_VALU_MFMA_BUSY_CYCLES  cycles a SIMD's matrix pipe is busy, summed over every SIMD (MI355X_MICROARCH.md: counts
#                               cycles, "= 32 x N_mfma for 32x32x16 bf16"; 16 per v_mfma_f32_16x16x32)
#     kernel cycles             GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs; same guide, "DVFS give-back")
#     SIMDs                     256 CUs x 4
#   An MFMA-bound kernel at the dense peak reads 1.0; achieved TFLOP/s / peak is the same quantity measured in time
#   (the two differ by the clock the chip holds under load: the fraction is in CYCLES).
# Usage: [AFX_WORKLOAD=xlsr_aasist] bash tools/pmc_mfma.sh <tag>  ->  gpurun_out/<tag>/pmc_mfma.json + pmc_mfma.txt
set -u
TAG=${1:-pmc_mfma}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  d=$(echo $c | tr ' ' '_' | cut -c1-60)
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$d" -- \
     python3 "$ROOT/tools/pmc_forward.py" > "$OUT/$d.log" 2>&1
  rc=$?
  echo "pass $c rc=$rc"
  if [ $rc -ne 0 ]; then tail -n 5 "$OUT/$d.log"; fi
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping"; exit $rc; fi
done
cd "$ROOT"
python3 - "$OUT" <<'PYEOF'
import collections, csv, glob, json, os, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if n.startswith("afx::"):
            agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
SIMDS = 256 * 4
res, lines = {}, []
for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", [0]))):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    rec = {"launches_sampled": len(d.get("GRBM_GUI_ACTIVE", d.get("SQ_LDS_IDX_ACTIVE", [])))}
    rec.update({c.lower(): v for c, v in m.items()})
    if m.get("GRBM_GUI_ACTIVE"):
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0
        rec["kernel_cycles"] = cyc
        rec["mfma_busy_frac"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * SIMDS)
    if m.get("SQ_LDS_IDX_ACTIVE"):
        rec["lds_conflict_per_active"] = m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"]
    if m.get("SQ_WAVE_CYCLES"):
        rec["wait_any_per_wave_cycle"] = m.get("SQ_WAIT_ANY", 0.0) / m["SQ_WAVE_CYCLES"]
    res[k] = rec
    lines.append(f"{k[:84]:84s} n={rec['launches_sampled']:4d}  mfma_busy_frac={rec.get('mfma_busy_frac', float('nan')):.3f}  "
                 f"lds_conflict/active={rec.get('lds_conflict_per_active', float('nan')):.3f}  wait_any/wave_cycle={rec.get('wait_any_per_wave_cycle', float('nan')):.3f}")
print("\n".join(lines))
open(out + "/pmc_mfma.txt", "w").write("\n".join(lines) + "\n")
stamp = {}
try:
    stamp = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(out))), "real-time-deepfake-speech-detection_amd", "lib", "build_stamp.json")))
except Exception:
    pass
json.dump({"git": stamp.get("git", "unknown"), "git_dirty": stamp.get("dirty"), "build_id": stamp.get("build_id", "unknown"), "dtype": os.environ.get("AFX_DTYPE", "fp16"), "note": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), mean per launch over 3 forwards of workload "
                   + os.environ.get("AFX_WORKLOAD", "conformer_student") + " (tools/pmc_forward.py); raw counter means beside it",
           "kernels": res}, open(out + "/pmc_mfma.json", "w"), indent=1)
PYEOF
find "$OUT" -name "*counter_collection.csv" -size +20M -delete 2>/dev/null
find "$OUT" -name "*kernel_trace.csv" -delete 2>/dev/null
exit 0
