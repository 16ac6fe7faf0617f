"""A/B of the row-complete conv tile's height (gemm_fit 14 / 13 / 12 = 128 / 96 / 64 rows, 1 = fitted) on the conv
layers of a batch (BENCH_B, default 16 = the teacher's): layers 1-6 of the feature extractor, interleaved rounds."""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import kernels as K  # noqa: E402
from afx._lib import check, lib  # noqa: E402

B = int(os.environ.get("BENCH_B", 16))
LAYERS = [(1, 12799, 3), (2, 6399, 3), (3, 3199, 3), (4, 1599, 3), (5, 799, 2), (6, 399, 2)]
# "fitted" = the default dispatch incl. the remainder split of multi-round layers; "fitted nosplit" switches that off
FITS = [("128 rows", 14), ("128 rows 4ph", 114), ("96 rows", 13), ("64 rows", 12), ("fitted", 1), ("fitted nosplit", 1001)]  # + 100: the 4-phase K-tile


def timeit(fn, reps=20):
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
    for li, Tin, k in LAYERS:
        x = torch.randn(B, Tin, 512, generator=g, device="cuda").half()
        wp = (torch.randn(512, k * 512, generator=g, device="cuda") * 0.03).half()
        bias = torch.randn(512, generator=g, device="cuda")
        ga = torch.ones(512, device="cuda")
        times = {n: [] for n, _ in FITS}
        for _ in range(5):
            for n, fit in FITS:
                check(lib().afx_debug_set(b"gemm_conv_split", 0 if fit >= 1000 else 1))
                check(lib().afx_debug_set(b"gemm_fit", fit % 100))
                check(lib().afx_debug_set(b"gemm_ph4", fit % 1000 // 100))
                times[n].append(timeit(lambda: K.conv_ln_act("fp16", x, wp, k, 2, bias, ga, bias)))
        check(lib().afx_debug_set(b"gemm_fit", 1))
        check(lib().afx_debug_set(b"gemm_ph4", 0))
        check(lib().afx_debug_set(b"gemm_conv_split", 1))
        Tout = (Tin - k) // 2 + 1
        print(f"layer {li} M={B * Tout:6d} K={k * 512}: " + "  ".join(f"{n}: {statistics.median(t):6.1f} us" for n, t in times.items()), flush=True)


if __name__ == "__main__":
    main()
