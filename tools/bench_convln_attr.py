"""Attribution of the fused conv + LayerNorm + GELU tile (8-phase 128x512) on conv layer 1 / 3 shapes:
timing-only gemm_nodma bits (8 no GELU, 32 no stores, 64 no epilogue at all).  Variants are interleaved over
several rounds after a long warm-up and the median is reported: measured one after the other from a cold
start, the first variant ("full") runs before the clocks have ramped and every saving is overstated."""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import kernels as K  # noqa: E402
from afx._lib import lib, check  # noqa: E402

VARIANTS = [("full", 0), ("no GELU", 8), ("no stores", 32), ("no GELU, no stores", 40), ("no epilogue", 64)]
g = torch.Generator(device="cuda").manual_seed(1)
for Tin in (12799, 3199):
    x = torch.randn(64, Tin, 512, generator=g, device="cuda").half()
    wp = (torch.randn(512, 1536, generator=g, device="cuda") * 0.03).half()
    bias = torch.randn(512, generator=g, device="cuda")
    ga = torch.ones(512, device="cuda")
    fn = lambda: K.conv_ln_act("fp16", x, wp, 3, 2, bias, ga, bias)
    for _ in range(40):
        fn()
    times = {name: [] for name, _ in VARIANTS}
    for _ in range(5):
        for name, bits in VARIANTS:
            check(lib().afx_debug_set(b"gemm_nodma", bits))
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) / 10 * 1e3)
    for name, _ in VARIANTS:
        print(f"Tin={Tin:6d} {name:22s} {statistics.median(times[name]):8.1f} us", flush=True)
# the same attribution for the dominant kernel class (8-phase 256x256 + round split): FC1 (GELU, operand-type
# output) and the out-projection (fp32 residual in and out)
for name, (M, N, K_), kw in (("fc1 12736x4096x1024 gelu", (12736, 4096, 1024), dict(act="gelu", out_f=False, out_h=True)),
                             ("qkv 12736x3072x1024", (12736, 3072, 1024), dict(out_f=False, out_h=True)),
                             ("out 12736x1024x1024 +resid", (12736, 1024, 1024), dict(out_f=True, out_h=False)),
                             ("fc2 12736x1024x4096 +resid", (12736, 1024, 4096), dict(out_f=True, out_h=False))):
    a = torch.randn(M, K_, generator=g, device="cuda").half()
    w = (torch.randn(N, K_, generator=g, device="cuda") * 0.03).half()
    bias = torch.randn(N, generator=g, device="cuda")
    resid = torch.randn(M, N, generator=g, device="cuda") if kw.get("out_f") else None
    fn = lambda: K.gemm("fp16", a, w, bias=bias, resid=resid, **kw)
    for _ in range(40):
        fn()
    times = {v: [] for v, _ in VARIANTS}
    for _ in range(5):
        for v, bits in VARIANTS:
            check(lib().afx_debug_set(b"gemm_nodma", bits))
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 10 * 1e3)
    print(f"{name:30s} " + "  ".join(f"{v}: {statistics.median(t):6.1f}" for v, t in times.items()) + "  (us)", flush=True)
check(lib().afx_debug_set(b"gemm_nodma", 0))
