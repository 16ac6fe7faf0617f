"""Diagnostic: the 4-wave form of the 256x256 tile (gemm4_kernel: one wave per SIMD, 128x128 outputs per wave, accumulators in
AGPRs; gemm_tile 4) against the 2-stage 128x128 tile -- bit for bit, over repeated launches -- before tools/bench_gemm.py
(BENCH_SET=w4) times it against the 8-wave kernel.  The kernel lives in the ATTRIBUTION build only:
    make -C real-time-deepfake-speech-detection_amd/csrc attr && AFX_LIB=real-time-deepfake-speech-detection_amd/lib/libafx_attr.so python tools/diag_gemm4.py"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import kernels as K  # noqa: E402
from afx._lib import check, lib  # noqa: E402

ok = True
for M, N, K_ in ((1000, 768, 64), (515, 512, 192), (300, 256, 128), (2048, 1024, 4096), (12736, 1024, 1024), (12736, 3072, 1024), (3184, 4096, 1024)):
    g = torch.Generator().manual_seed(K_ + M)
    A = torch.randn(M, K_, generator=g).half().cuda()
    W = (torch.randn(N, K_, generator=g) / math.sqrt(K_)).half().cuda()
    bias = torch.randn(N, generator=g).cuda()
    R = torch.randn(M, N, generator=g).cuda()
    check(lib().afx_debug_set(b"gemm_tile", 0))
    want_f, want_h = K.gemm("fp16", A, W, bias=bias, act="gelu", resid=R, out_f=True, out_h=True)
    for tile, name in ((4, "4-wave tile (two K-tile buffers)"), (10, "4-wave tile (5-stage ring)")):
        check(lib().afx_debug_set(b"gemm_tile", tile))
        same = True
        for _ in range(8):
            got_f, got_h = K.gemm("fp16", A, W, bias=bias, act="gelu", resid=R, out_f=True, out_h=True)
            same = same and torch.equal(got_f, want_f) and torch.equal(got_h, want_h)
        check(lib().afx_debug_set(b"gemm_tile", -1))
        print(f"M {M} N {N} K {K_}: {name} bit-identical to the 2-stage tile over 8 launches: {same}" + ("" if same else f"  max |d| {(got_f - want_f).abs().max().item():.3e}"), flush=True)
        ok = ok and same
sys.exit(0 if ok else 1)
