"""How much of a GEMM's time is operand delivery from beyond the L2: the same launch with A (and W) replaced by ONE row
repeated (row stride 0: every LDS-DMA is a cache hit), against the real operands.  Student shapes, 8-wave kernel."""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import kernels as K  # noqa: E402


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
    M = int(os.environ.get("BENCH_M", 12736))
    for name, N, Kk, resid in (("qkv", 3072, 1024, False), ("out", 1024, 1024, True), ("fc1", 4096, 1024, False), ("fc2", 1024, 4096, True)):
        a = torch.randn(M, Kk, generator=g, device="cuda").half()
        w = (torch.randn(N, Kk, generator=g, device="cuda") * 0.03).half()
        bias = torch.randn(N, generator=g, device="cuda")
        x = torch.randn(M, N, generator=g, device="cuda") if resid else None
        variants = {"real operands": (a, w), "A = one row": (a[:1].expand(M, Kk), w), "A and W = one row each": (a[:1].expand(M, Kk), w[:1].expand(N, Kk))}
        times = {k: [] for k in variants}
        for _ in range(5):
            for k, (aa, ww) in variants.items():
                if resid:
                    f = lambda: K.gemm("fp16", aa, ww, bias=bias, resid=x, out_f=True, out_h=False)
                else:
                    f = lambda: K.gemm("fp16", aa, ww, bias=bias, act="gelu" if name == "fc1" else None, out_f=False, out_h=True)
                times[k].append(timeit(f))
        print(f"{name} M={M} N={N} K={Kk}: " + "  ".join(f"{k}: {statistics.median(t):6.1f} us" for k, t in times.items()), flush=True)


if __name__ == "__main__":
    main()
