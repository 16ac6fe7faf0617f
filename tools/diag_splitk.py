"""Diagnostic: what a split-K form of the teacher's FFN2 (M = 16 x 199, N = 1024, K = 4096) would cost with today's kernels:
the 256x256 kernel over 4 K-quarters writing fp32 partials (= FFN1's shape with an fp32 output) + a pass that sums them."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import kernels  # noqa: E402


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    M = 3184
    A4 = (0.1 * torch.randn(M, 4096, device="cuda")).half()
    W2 = (0.03 * torch.randn(1024, 4096, device="cuda")).half()
    A1 = (0.1 * torch.randn(M, 1024, device="cuda")).half()
    W1 = (0.03 * torch.randn(4096, 1024, device="cuda")).half()
    b1, b4 = torch.zeros(1024, device="cuda"), torch.zeros(4096, device="cuda")
    resid = torch.randn(M, 1024, device="cuda")
    print("ffn2 as today (fp32 out + resid):      %.1f us" % timeit(lambda: kernels.gemm("fp16", A4, W2, bias=b1, resid=resid, out_f=True, out_h=False)))
    print("ffn1 shape, half out:                  %.1f us" % timeit(lambda: kernels.gemm("fp16", A1, W1, bias=b4, out_f=False, out_h=True)))
    print("ffn1 shape, fp32 out (= the partials): %.1f us" % timeit(lambda: kernels.gemm("fp16", A1, W1, out_f=True, out_h=False)))
    part = torch.randn(M, 4, 1024, device="cuda")
    out = torch.empty(M, 1024, device="cuda")
    print("torch sum of 4 partials alone:         %.1f us" % timeit(lambda: torch.sum(part, 1, out=out)))

    def both():
        of, _ = kernels.gemm("fp16", A1, W1, out_f=True, out_h=False)
        torch.sum(of.view(M, 4, 1024), 1, out=out)
    print("partials GEMM + torch sum back to back: %.1f us (includes one torch.empty per call)" % timeit(both))
    x = torch.randn(M, 1024, device="cuda")
    print("layer_norm(fp32 3184 x 1024) torch:    %.1f us" % timeit(lambda: torch.nn.functional.layer_norm(x, (1024,))))


if __name__ == "__main__":
    main()
