"""A/B of the two-chain tile family (gemm_k2_kernel: two K-halves in one workgroup, reduced through LDS) against the
single-chain tiles on the long-K / short-N products at small row counts (the teacher's out-proj and FC2 at B = 16 and
smaller batches), interleaved rounds in one process; plus agreement of the two forms (fp32 rounding of one final add)
and bit-identity of the 8-wave (tile 9) and 4-wave (tile 10) members of the family."""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import kernels as K  # noqa: E402
from afx._lib import check, lib  # noqa: E402


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
    for B in (16, 8, 4, 1, 24):
        M = B * 199
        for name, N, Kk in (("out", 1024, 1024), ("fc2", 1024, 4096)):
            a = torch.randn(M, Kk, generator=g, device="cuda").half()
            w = (torch.randn(N, Kk, generator=g, device="cuda") * 0.03).half()
            bias = torch.randn(N, generator=g, device="cuda")
            x = torch.randn(M, N, generator=g, device="cuda")
            run = lambda: K.gemm("fp16", a, w, bias=bias, resid=x, out_f=True, out_h=False)[0]
            times, outs = {0: [], 1: []}, {}
            for _ in range(5):
                for k2 in (0, 1):
                    check(lib().afx_debug_set(b"gemm_k2", k2))
                    times[k2].append(timeit(run))
                    outs[k2] = run()
            check(lib().afx_debug_set(b"gemm_k2", 1))
            d = (outs[0] - outs[1]).abs().max().item()
            fl = 2.0 * M * N * Kk
            print(f"{name} M={M:5d} (B={B:2d}) K={Kk}: single chain {statistics.median(times[0]):6.1f} us ({fl / statistics.median(times[0]) / 1e6:4.0f} TF)  "
                  f"two chains {statistics.median(times[1]):6.1f} us ({fl / statistics.median(times[1]) / 1e6:4.0f} TF)  max |diff| {d:.1e}", flush=True)


if __name__ == "__main__":
    main()
