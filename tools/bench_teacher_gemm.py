"""A/B of GEMM tile instances (gemm_tile override) on the teacher's (B = 16, M = 3184) products, interleaved rounds in
one process.  (Round 2 also measured here, and rejected: split-K with fp32 partial planes folded into the next LayerNorm,
and 8-wave 128x128 / 128x256 / 256x128 2-stage tiles -- profiles/r02_teacher_gemm_tile_splitk_ab.txt.)"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import kernels as K  # noqa: E402
from afx._lib import check, lib  # noqa: E402

M = int(os.environ.get("BENCH_M", 3184))
SHAPES = [("qkv", 3072, 1024, False), ("out", 1024, 1024, True), ("fc1", 4096, 1024, False), ("fc2", 1024, 4096, True)]
# (name, gemm_tile, gemm_fit): the 8-phase kernel at forced heights 256 / 224 / 192 / 160 rows and fitted
# gemm_fit + 100 = the 4-phase K-tile (full-height tiles only), + 200 = the two-buffer two-phase form instead of the three-buffer A ring
TILES = [("auto", -1, 1), ("auto 4ph", -1, 101), ("auto 2buf", -1, 201), ("auto nofit", -1, 0), ("128x128/4w", 0, 1), ("128x64/4w", 5, 1), ("8ph 256", 3, 8), ("8ph 256 4ph", 3, 108), ("8ph 256 2buf", 3, 208), ("8ph 224 2buf", 3, 207), ("8ph 160 2buf", 3, 205), ("8ph 224", 3, 7),
         ("8ph 192", 3, 6), ("8ph 160", 3, 5)]


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
    for name, N, Kk, resid in SHAPES:
        a = torch.randn(M, Kk, generator=g, device="cuda").half()
        w = (torch.randn(N, Kk, generator=g, device="cuda") * 0.03).half()
        bias = torch.randn(N, generator=g, device="cuda")
        x = torch.randn(M, N, generator=g, device="cuda") if resid else None
        ga = torch.ones(N, device="cuda")
        flops = 2.0 * M * N * Kk
        variants = {}
        for tn, tv, fit in TILES:
            def f(tv=tv, fit=fit):
                check(lib().afx_debug_set(b"gemm_tile", tv))
                check(lib().afx_debug_set(b"gemm_fit", fit % 100))
                check(lib().afx_debug_set(b"gemm_ph4", fit // 100))
                if resid:
                    K.gemm("fp16", a, w, bias=bias, resid=x, out_f=True, out_h=False)
                else:
                    K.gemm("fp16", a, w, bias=bias, act="gelu" if name == "fc1" else None, out_f=False, out_h=True)
            variants[tn] = f
        if resid:
            variants["LN alone"] = lambda: K.rownorm("fp16", x, ga, bias, out_f=False, out_h=True)
        times = {k: [] for k in variants}
        for _ in range(5):
            for k, f in variants.items():
                times[k].append(timeit(f))
        check(lib().afx_debug_set(b"gemm_tile", -1))
        check(lib().afx_debug_set(b"gemm_fit", 1))
        check(lib().afx_debug_set(b"gemm_ph4", 0))
        print(f"{name} M={M} N={N} K={Kk}: " + "  ".join(
            f"{k}: {statistics.median(t):6.1f} us" + (f" ({flops / statistics.median(t) / 1e6:5.0f} TF)" if "LN" not in k else "")
            for k, t in times.items()), flush=True)


if __name__ == "__main__":
    main()
