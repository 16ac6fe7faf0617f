"""Fixed cost per output tile of the 8-phase GEMM: time against K at fixed M x N (fc1 shape)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import kernels as K  # noqa: E402
from afx._lib import lib, check  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(1)
for (M, N) in [(12736, 4096), (12736, 1024), (12800, 4096), (65536, 512)]:
    for act, out_f in [("gelu", False), (None, False), (None, True)]:
        row = []
        for Kk in [64, 256, 512, 1024, 2048, 4096]:
            a = torch.randn(M, Kk, generator=g, device="cuda").half()
            w = (torch.randn(N, Kk, generator=g, device="cuda") * 0.03).half()
            bias = torch.randn(N, generator=g, device="cuda")
            check(lib().afx_debug_set(b"gemm_tile", 3))
            fn = lambda: K.gemm("fp16", a, w, bias=bias, act=act, out_f=out_f, out_h=not out_f)
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            row.append(f"K={Kk}: {e0.elapsed_time(e1) / 20 * 1e3:6.1f} us")
        print(f"M={M} N={N} act={act} fp32out={out_f}:  " + "  ".join(row), flush=True)
check(lib().afx_debug_set(b"gemm_tile", -1))
