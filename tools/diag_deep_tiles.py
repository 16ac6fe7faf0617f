"""Diagnostic: gemm_deep_kernel (deep pipeline + pre-read, one workgroup per CU) against the shipped selection on the teacher's
products: time and bit-identity.  gemm_tile: -1 default, 5 = 128x64 two-stage (the shipped choice at M = 16 x 199),
6 = 128x128 / 4 waves / 4 buffers, 7 = 128x128 / 8 waves / 4 buffers, 8 = 128x64 / 4 waves / 3 buffers, 9 = 128x128 / 4 waves / 3 buffers."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import kernels  # noqa: E402
from afx._lib import lib  # noqa: E402


def main():
    Ms = [int(a) for a in sys.argv[1:]] or [3184]
    # warm the clocks
    x = torch.randn(4096, 4096, device="cuda")
    for _ in range(20):
        x @ x
    for M in Ms:
        for name, N, K in (("out-proj", 1024, 1024), ("ffn2", 1024, 4096), ("qkv", 3072, 1024), ("ffn1", 4096, 1024)):
            A = (0.1 * torch.randn(M, K, device="cuda")).half()
            W = (0.03 * torch.randn(N, K, device="cuda")).half()
            b = 0.1 * torch.randn(N, device="cuda")
            resid = torch.randn(M, N, device="cuda")
            ref = None
            line = f"M {M:5d} {name:9s}"
            for tile in (-1, 5, 6, 7, 8, 9):
                lib().afx_debug_set(b"gemm_tile", tile)
                fn = lambda: kernels.gemm("fp16", A, W, bias=b, resid=resid, out_f=True, out_h=True)
                for _ in range(3):
                    of, oh = fn()
                torch.cuda.synchronize()
                if ref is None:
                    ref = (of.clone(), oh.clone())
                same = torch.equal(of, ref[0]) and torch.equal(oh, ref[1])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(30):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / 30 * 1e3
                line += f" | {tile:2d}: {us:6.1f}{'' if same else ' DIFF'}"
            print(line, flush=True)
    lib().afx_debug_set(b"gemm_tile", -1)


def maps():
    """The deep 128x64 tile under the workgroup -> tile orders 0 / 1 / 2 (default)."""
    for M in (3184,):
        for name, N, K in (("out-proj", 1024, 1024), ("ffn2", 1024, 4096)):
            A = (0.1 * torch.randn(M, K, device="cuda")).half()
            W = (0.03 * torch.randn(N, K, device="cuda")).half()
            b = 0.1 * torch.randn(N, device="cuda")
            resid = torch.randn(M, N, device="cuda")
            line = f"M {M:5d} {name:9s}"
            ref = None
            for mp in (2, 0, 1, 2):
                lib().afx_debug_set(b"gemm_map", mp)
                fn = lambda: kernels.gemm("fp16", A, W, bias=b, resid=resid, out_f=True, out_h=True)
                for _ in range(5):
                    of, oh = fn()
                torch.cuda.synchronize()
                ref = (of.clone(), oh.clone()) if ref is None else ref
                same = torch.equal(of, ref[0]) and torch.equal(oh, ref[1])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(30):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                line += f" | map {mp}: {e0.elapsed_time(e1) / 30 * 1e3:6.1f}{'' if same else ' DIFF'}"
            print(line, flush=True)
    lib().afx_debug_set(b"gemm_map", -1)


if __name__ == "__main__":
    maps() if "maps" in sys.argv else main()
