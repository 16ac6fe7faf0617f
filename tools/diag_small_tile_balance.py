"""Diagnostic: the 128x64 four-wave tile on the teacher's N = 1024 products as a function of how evenly its tiles fall on the
256 CUs (M = 16 x 199 = 3184 gives 400 tiles: 144 CUs hold two, 112 hold one)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import kernels  # noqa: E402
from afx._lib import lib  # noqa: E402


def main():
    lib().afx_debug_set(b"gemm_tile", 5)  # force the 128x64 four-wave instance
    for name, N, K in (("out-proj", 1024, 1024), ("ffn2", 1024, 4096)):
        for M in (2048, 3072, 3184, 3200, 3328, 4096, 6144, 8192):
            A = (0.1 * torch.randn(M, K, device="cuda")).half()
            W = (0.03 * torch.randn(N, K, device="cuda")).half()
            b = torch.zeros(N, device="cuda")
            for _ in range(5):
                kernels.gemm("fp16", A, W, bias=b, out_f=False, out_h=True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                kernels.gemm("fp16", A, W, bias=b, out_f=False, out_h=True)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 50 * 1e3
            tiles = ((M + 127) // 128) * (N // 64)
            print(f"{name:9s} M {M:5d}: {tiles:4d} tiles ({tiles / 256:4.2f} per CU) {us:7.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
