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
    gemm4) export AFX_LIB=$PWD/real-time-deepfake-speech-detection_amd/lib/libafx_attr.so; timeout -k 10 300 python tools/diag_gemm4.py > $O/gemm4_identity.txt 2>&1; rc=$?; cat $O/gemm4_identity.txt | grep -v amdgpu.ids; if [ $rc -ne 0 ]; then echo "gemm4 identity failed (rc $rc): no timing"; else BENCH_SET=w4 timeout -k 10 600 python tools/bench_gemm.py > $O/gemm4_ab.txt 2>&1; grep -v amdgpu.ids $O/gemm4_ab.txt | cut -c1-300; fi;;
    s3chain) timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 300 -s -k "split_precision_conformer_chains or fp32_conformer_student or ragged_batch_in_split or conformer_student_scores or myconformer or conformer_head_alone" > $O/pytest_s3chain.log 2>&1; rc=$?; grep -v "amdgpu.ids\|RuntimeWarning\|check(" $O/pytest_s3chain.log | tail -12 | cut -c1-300; if [ $rc -ne 0 ]; then echo "tests failed (rc $rc): no timing"; else timeout -k 10 300 python tools/diag_s3_chain.py > $O/s3_chain.txt 2>&1; grep -v "amdgpu.ids\|RuntimeWarning\|check(" $O/s3_chain.txt | cut -c1-400; fi;;
    kvchunk) timeout -k 10 900 python -m pytest tests/test_gpu_streaming_kv.py -q --timeout 600 > $O/pytest_kv.log 2>&1; rc=$?; tail -4 $O/pytest_kv.log | cut -c1-300; if [ $rc -ne 0 ]; then echo "tests failed (rc $rc): no timing"; else timeout -k 10 500 python tools/stream_bench.py --workload conformer_student --modes kv-cached > $O/rtf_student.txt 2>&1; grep -v "amdgpu.ids\|RuntimeWarning\|check(" $O/rtf_student.txt | cut -c1-300; timeout -k 10 500 python tools/stream_bench.py --workload xlsr_aasist --streams 1 64 512 --modes kv-cached > $O/rtf_teacher.txt 2>&1; grep -v "amdgpu.ids\|RuntimeWarning\|check(" $O/rtf_teacher.txt | cut -c1-300; fi;;
    s3small) timeout -k 10 400 python tools/diag_s3_small.py > $O/s3_small.txt 2>&1; grep -v "amdgpu.ids\|RuntimeWarning\|check(" $O/s3_small.txt | cut -c1-300;;
    vendor) BENCH_SET=vendor timeout -k 10 400 python tools/bench_gemm.py > $O/vendor_gemm.txt 2>&1; grep -v "amdgpu.ids\|RuntimeWarning\|check(" $O/vendor_gemm.txt | cut -c1-300;;
    pkprobe) /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/pk_hazard_probe.hip -o /tmp/pk_hazard_probe > $O/pk_probe_build.log 2>&1 && timeout -k 10 300 /tmp/pk_hazard_probe > $O/pk_hazard_probe.txt 2>&1; cat $O/pk_hazard_probe.txt | cut -c1-260; python tools/scan_pk_hazard.py | tee $O/scan_pk.txt; timeout -k 10 600 python -m pytest tests/test_gpu_aasist.py -q --timeout 300 -k "valu_conv0_kernels or overlaps_the_backend" > $O/pytest_pk.log 2>&1; tail -5 $O/pytest_pk.log | cut -c1-300;;
    c0pk) timeout -k 10 300 python tools/diag_conv0_pk.py 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|check(" | cut -c1-300 | tee $O/conv0_pk_product.txt; AFX_LIB=$PWD/real-time-deepfake-speech-detection_amd/lib/libafx_c0pk.so timeout -k 10 300 python tools/diag_conv0_pk.py 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|check(" | cut -c1-300 | tee $O/conv0_pk_packed.txt; AFX_LIB=$PWD/real-time-deepfake-speech-detection_amd/lib/libafx_c0pkx.so timeout -k 10 300 python tools/diag_conv0_pk.py 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|check(" | cut -c1-300 | tee $O/conv0_pk_packed_not_in_place.txt;;
    nopkbench) for rep in 1 2; do for vn in ${VARIANTS:-product _allpk _nopkall}; do v=$vn; [ "$vn" = product ] && v=""; L=$PWD/real-time-deepfake-speech-detection_amd/lib/libafx$v.so; AFX_LIB=$L timeout -k 10 300 python bench.py --cpu-sample 0 --steps 30 --warmup 5 2> $O/nopk_bench.err | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['config3']; k=d.get('contract',{}); print('lib', '$v' or 'product', 'student', d['value'], d['ms_per_step'], 'one/two', d['issue_probe'].get('one_stream_ms_per_step'), d['issue_probe'].get('two_stream_ms_per_step'), '| teacher', c['value'], c['ms_per_step'], '| contract', k.get('value'), c.get('contract',{}).get('value'))" | tee -a $O/nopk_bench.txt; done; done;;
    pkunits) timeout -k 10 500 python tools/diag_pk_units.py 40 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|check(" | cut -c1-200 | tee $O/pk_units.txt;;
    vendornames) R=$PWD; (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/vn -- python3 $R/tools/diag_vendor_kernel_names.py > $R/$O/vn.log 2>&1); python3 - $O <<'PYEOF'
import csv, glob, sys, collections
seen = collections.OrderedDict()
for f in glob.glob(sys.argv[1] + "/vn/**/*kernel_trace.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    for r in rows:
        n = r["Kernel_Name"]
        if n.startswith("Cijk") or "gemm" in n.lower() or "Custom_" in n:
            k = (n, r["Grid_Size_X"], r["Workgroup_Size_X"], r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "")), r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""))
            seen.setdefault(k, [0, 0]); seen[k][0] += 1; seen[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for (n, gx, wg, lds, vg, ag), (c, t) in seen.items():
    print(f"x{c} {t / c / 1e3:8.1f} us  grid {gx} wg {wg} lds {lds} vgpr {vg} agpr {ag}  {n[:200]}")
PYEOF
    ;;
    vtr) for rep in 1 2; do timeout -k 10 200 python tools/diag_mhsa_split_vtr.py 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|check(" | cut -c1-200 | tee -a $O/vtr.txt; AFX_LIB=$PWD/real-time-deepfake-speech-detection_amd/lib/libafx_vtr.so timeout -k 10 200 python tools/diag_mhsa_split_vtr.py 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|check(" | cut -c1-200 | tee -a $O/vtr.txt; done;;
    capipe) for rep in 1 2; do timeout -k 10 200 python tools/diag_conf_attn_pipe.py 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|check(" | cut -c1-200 | tee -a $O/capipe.txt; AFX_LIB=$PWD/real-time-deepfake-speech-detection_amd/lib/libafx_ca.so timeout -k 10 200 python tools/diag_conf_attn_pipe.py 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|check(" | cut -c1-200 | tee -a $O/capipe.txt; done;;
    newtests) timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout 600 -k "deep_tile or overlaps_the_backend or test_gpu_bench or outlier or per_engine or forward_hooks or full_depth" > $O/pytest_new.log 2>&1; tail -8 $O/pytest_new.log | cut -c1-300;;
    stale) timeout -k 10 300 python tools/diag_s3_stale.py > $O/s3_stale.txt 2>&1; grep -v "amdgpu.ids\|RuntimeWarning\|check(" $O/s3_stale.txt | cut -c1-300;;
    headrace) timeout -k 10 300 python tools/diag_head_race.py > $O/head_race.txt 2>&1; grep -v "amdgpu.ids\|RuntimeWarning\|check(" $O/head_race.txt | cut -c1-300;;
    guard) timeout -k 10 300 python tools/diag_ws_guard.py > $O/ws_guard.txt 2>&1; grep -v "amdgpu.ids\|RuntimeWarning\|check(" $O/ws_guard.txt | cut -c1-300;;
    bisect) for k in 1 2 3 5 8 9 10 0; do timeout -k 10 100 python tools/diag_two_stream_which.py fp16x3 $k 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|check(\|   batch" | cut -c1-260 | tee -a $O/bisect.txt; done;;
    which1) timeout -k 10 300 python tools/diag_two_stream_which.py fp16x3 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|check(\|workspace bytes differ" | cut -c1-330 | tee $O/which1.txt;;
    which) for d in fp16x3 fp16; do timeout -k 10 200 python tools/diag_two_stream_which.py $d 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|check(" | cut -c1-400 | tee -a $O/two_stream_which.txt; done;;
    conv0race) timeout -k 10 200 python tools/diag_conv0_race.py 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|check(" | cut -c1-300 | tee $O/conv0_race.txt;;
    *) echo "unknown step $step";;
  esac
done
