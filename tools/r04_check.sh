# One GPU-box visit of round 4 (scratch driver; outputs under gpurun_out/<tag>): usage bash tools/r04_check.sh <tag> <steps...>
set -u
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
for step in "$@"; do
  case $step in
    tests) timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest.log 2>&1; rc=$?; tail -4 $O/pytest.log; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest killed"; exit $rc; fi;;
    bench) timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err; rc=$?; echo "bench rc $rc"; tail -3 $O/bench.err; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
           python - $O/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
def show(name, r):
    print(name, r["value"], "utt/s", r["ms_per_step"], "ms", r.get("dtype"), "frac", r["roofline"]["frac"], r["roofline"]["kernel"], r.get("issue_probe"), "parity", {k: v for k, v in (r.get("parity") or {}).items() if k in ("max_abs_dlogit_vs_oracle", "utterances_keeping_every_topk_decision", "max_where_topk_kept", "max_where_topk_changed")}, r.get("parity_ok"))
show("headline", d)
if "contract" in d: show("  contract", d["contract"])
if "config3" in d:
    show("config3", d["config3"])
    if "contract" in d["config3"]: show("  contract", d["config3"]["contract"])
print("with_pcie", d.get("with_pcie"))
PY
           ;;
    outliers) timeout -k 10 600 python tools/diag_outliers.py > $O/outliers.txt 2>&1; cat $O/outliers.txt | cut -c1-400;;
    s3tiles) timeout -k 10 400 python tools/diag_s3_tiles.py > $O/s3_tiles.txt 2>&1; cut -c1-500 $O/s3_tiles.txt;;
    s3knobs) timeout -k 10 400 python tools/diag_s3_knobs.py > $O/s3_knobs.txt 2>&1; cut -c1-500 $O/s3_knobs.txt;;
    *) echo "unknown step $step";;
  esac
done
