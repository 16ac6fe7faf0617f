"""Diagnostic: the trunk's GEMM shapes one at a time through the unit entry point (afx_k_gemm, half out, bias): the
selection the engine makes for each (M, N, K) and the rate it reaches.  python tools/diag_gemm_shapes.py [M ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import kernels  # noqa: E402


def main():
    E = 768 if "--student" in sys.argv else 1024  # XLS-R trunk width: 1024; (the student keeps 1024 too -- 768 only as a what-if)
    Ms = [int(a) for a in sys.argv[1:] if a.isdigit()] or [3184]
    for M in Ms:
        for name, N, K in (("qkv", 3 * E, E), ("out-proj", E, E), ("ffn1", 4 * E, E), ("ffn2", E, 4 * E)):
            A = (0.1 * torch.randn(M, K, device="cuda")).half()
            W = (0.03 * torch.randn(N, K, device="cuda")).half()
            b = torch.zeros(N, device="cuda")
            for _ in range(5):
                kernels.gemm("fp16", A, W, bias=b, out_f=False, out_h=True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            n = 50
            for _ in range(n):
                kernels.gemm("fp16", A, W, bias=b, out_f=False, out_h=True)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / n * 1e3
            print(f"M {M:5d} {name:9s} N {N:5d} K {K:5d}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
