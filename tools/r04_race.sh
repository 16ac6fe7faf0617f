#!/usr/bin/env bash
# Two-stream defect, bisected on the build that showed it (git worktree _old = 26d8841, conv0_kernel<F32T> in the fp16x3 path):
# (set-up, not part of the repository's tree: git worktree add _old 26d8841 && (cd _old && git apply ../tools/r04_race_variants.patch) &&
#  make -C _old/real-time-deepfake-speech-detection_amd/csrc && for v in red1:-DC0_RED=1 red2:-DC0_RED=2 nopk:-DC0_NOPK=1; do
#  make -C _old/real-time-deepfake-speech-detection_amd/csrc variant NAME=${v%%:*} DEFS=${v#*:}; done)
# variants of that kernel (reduction by ds_bpermute; s_nop around the permlane swaps; no packed fp32 math) and of what runs beside the
# trunk (the back-end's first kernel; any small kernel; nothing).  usage: bash tools/r04_race.sh <tag>
set -u
O=$PWD/gpurun_out/${1:-r04_race}; mkdir -p $O
cd _old || exit 1
export DIAG_PASSES=6 DIAG_BRIEF=1
L=$PWD/real-time-deepfake-speech-detection_amd/lib
run() {  # name, lib suffix, DIAG_SIDE
  echo "== $1" | tee -a $O/race.txt
  AFX_LIB=$L/libafx$2.so DIAG_SIDE=$3 timeout -k 10 240 python tools/diag_two_stream_which.py fp16x3 1 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|check(\|workspace bytes differ from" | cut -c1-400 | tee -a $O/race.txt
  rc=${PIPESTATUS[0]}; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed (rc $rc): stopping"; exit $rc; fi
}
if [ "${2:-}" = "aggressor" ]; then
  run "baseline kernel, chip-filling elementwise kernel beside the trunk (VALU + memory, no matrix instruction)" "" valu
  run "baseline kernel, vendor fp16 matrix-core GEMM beside the trunk" "" mm16
  run "baseline kernel, vendor fp32 GEMM beside the trunk" "" mm32
  run "baseline kernel, the back-end's first kernel beside the trunk (again)" "" ""
  run "conv0 without packed fp32 math, the back-end's first kernel beside the trunk (again)" _nopk ""
  run "conv0 without packed fp32 math, vendor fp16 matrix-core GEMM beside the trunk" _nopk mm16
  exit 0
fi
run "baseline (build 26d8841), back-end's first kernel beside the trunk" "" ""
run "conv0 reduction by ds_bpermute (no DPP, no permlane swap)" _red1 ""
run "conv0 reduction with s_nop 7 around the permlane swaps" _red2 ""
run "conv0 without packed fp32 math (scalar v_fma chain)" _nopk ""
run "baseline kernel, ANY small kernel (torch add_) beside the trunk" "" torch
run "baseline kernel, nothing beside the trunk (event + wait only)" "" none
