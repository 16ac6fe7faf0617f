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
check(lib().afx_debug_set(b"gemm_nodma", 0))
