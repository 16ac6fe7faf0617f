"""Fixed cost per output tile of the 8-phase GEMM: time against K at fixed M x N, with parts of the
epilogue switched off through the timing-only gemm_nodma bits (8 no activation, 32 no stores, 64 no epilogue)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import kernels as K  # noqa: E402
from afx._lib import lib, check  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(1)
check(lib().afx_debug_set(b"gemm_tile", 3))
for (M, N) in [(12736, 4096), (12736, 1024)]:
    for name, bits in [("full (bias+gelu, fp16 out)", 0), ("no activation", 8), ("no stores", 32), ("no act, no stores", 40), ("no epilogue", 64)]:
        row = []
        for Kk in [64, 512, 1024, 4096]:
            a = torch.randn(M, Kk, generator=g, device="cuda").half()
            w = (torch.randn(N, Kk, generator=g, device="cuda") * 0.03).half()
            bias = torch.randn(N, generator=g, device="cuda")
            check(lib().afx_debug_set(b"gemm_nodma", bits))
            fn = lambda: K.gemm("fp16", a, w, bias=bias, act="gelu", out_f=False, out_h=True)
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            row.append(f"K={Kk}: {e0.elapsed_time(e1) / 20 * 1e3:6.1f} us")
        print(f"M={M} N={N} {name:28s} " + "  ".join(row), flush=True)
check(lib().afx_debug_set(b"gemm_nodma", 0))
check(lib().afx_debug_set(b"gemm_tile", -1))
