"""What bounds the K-loop of the 8-wave GEMM kernels, on the model's own shapes (epilogue off): parts of the loop are
switched off through the timing-only gemm_nodma bits of the attribution build (make attr; AFX_LIB=.../libafx_attr.so):
128 / 256 / 512 / 1024 = one operand half-tile's LDS-DMA off, 2048 = no MFMAs, 4096 = no LDS fragment reads; gemm_ph4 = 1
selects the 4-phase K-tile (16-MFMA segments) instead of the two phases of 32.  K-tiles per workgroup are printed so
that the times read as us per K-tile."""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import kernels as K  # noqa: E402
from afx._lib import check, lib  # noqa: E402

ALL_DMA = 128 + 256 + 512 + 1024
VARIANTS = [("full loop", 0), ("no DMA at all", ALL_DMA), ("no MFMA", 2048), ("no LDS reads", 4096), ("DMA only", 2048 + 4096),
            ("MFMA only (+ barriers)", ALL_DMA + 4096), ("reads only", ALL_DMA + 2048)]


def timeit(fn, reps=10):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    g = torch.Generator(device="cuda").manual_seed(1)
    check(lib().afx_debug_set(b"gemm_tile", 3))
    check(lib().afx_debug_set(b"gemm_fit", 8))
    cases = []
    for name, M, N, Kk in (("fc1 12288x4096x1024 (3 rounds)", 12288, 4096, 1024), ("fc2 12736x1024x4096 (1 round)", 12736, 1024, 4096)):
        a = torch.randn(M, Kk, generator=g, device="cuda").half()
        w = (torch.randn(N, Kk, generator=g, device="cuda") * 0.03).half()
        bias = torch.randn(N, generator=g, device="cuda")
        rounds = -(-((M + 255) // 256) * (N // 256) // 256)
        cases.append((name, rounds * Kk // 64, lambda a=a, w=w, bias=bias: K.gemm("fp16", a, w, bias=bias, out_f=False, out_h=True)))
    x = torch.randn(64, 12799, 512, generator=g, device="cuda").half()
    wp = (torch.randn(512, 1536, generator=g, device="cuda") * 0.03).half()
    bias = torch.randn(512, generator=g, device="cuda")
    ga = torch.ones(512, device="cuda")
    cases.append(("conv layer 1 (12.5 rounds)", 13 * 24, lambda: K.conv_ln_act("fp16", x, wp, 3, 2, bias, ga, bias)))
    for name, ktiles, f in cases:
        for _ in range(20):
            f()
        for ph4 in (0, 1):
            check(lib().afx_debug_set(b"gemm_ph4", ph4))
            times = {n: [] for n, _ in VARIANTS}
            for _ in range(5):
                for n, bits in VARIANTS:
                    check(lib().afx_debug_set(b"gemm_nodma", 64 | bits))
                    times[n].append(timeit(f))
            check(lib().afx_debug_set(b"gemm_nodma", 0))
            full = timeit(f)
            print(f"{name:32s} {'4-phase' if ph4 else '2-phase'} with epilogue {full:7.1f} us | " + "  ".join(
                f"{n}: {statistics.median(t):6.1f} ({statistics.median(t) / ktiles:5.3f}/kt)" for n, t in times.items()), flush=True)
    check(lib().afx_debug_set(b"gemm_ph4", 0))
    check(lib().afx_debug_set(b"gemm_tile", -1))
    check(lib().afx_debug_set(b"gemm_fit", 1))


if __name__ == "__main__":
    main()
