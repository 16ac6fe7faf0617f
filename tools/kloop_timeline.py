"""Clock-stamp timeline of the two-phase K-loop (attribution build: AFX_LIB=.../libafx_attr.so, gemm_nodma bit 8192 selects
the stamping instance of the 256x256 kernel): per wave row, mean cycles of the four segments of each phase over the K-tiles
of workgroup 0's first output tile --
    read part   = start of phase -> own DMA landed and own LDS reads retired
    wait 1      = -> first barrier passed (the other row finishing its MFMA burst)
    mfma        = -> 32 MFMAs issued
    wait 2      = -> second barrier passed (the other row finishing its read part)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import kernels as K  # noqa: E402
from afx._lib import check, lib  # noqa: E402


def main():
    g = torch.Generator(device="cuda").manual_seed(1)
    check(lib().afx_debug_set(b"gemm_tile", 3))
    check(lib().afx_debug_set(b"gemm_fit", 8))
    ALL_DMA = 128 + 256 + 512 + 1024
    for name, M, N, Kk, bits in (("fc1 12288x4096x1024", 12288, 4096, 1024, 0), ("fc1, no DMA", 12288, 4096, 1024, ALL_DMA), ("fc1, no LDS reads", 12288, 4096, 1024, 4096),
                                 ("fc1, no MFMA", 12288, 4096, 1024, 2048), ("fc1, no DMA, no reads", 12288, 4096, 1024, ALL_DMA + 4096),
                                 ("out 12736x1024x1024", 12736, 1024, 1024, 0), ("fc2 12736x1024x4096 (first 16 K-tiles)", 12736, 1024, 4096, 0)):
        a = torch.randn(M, Kk, generator=g, device="cuda").half()
        w = (torch.randn(N, Kk, generator=g, device="cuda") * 0.03).half()
        bias = torch.randn(N, generator=g, device="cuda")
        check(lib().afx_debug_set(b"gemm_nodma", 0))
        for _ in range(5):
            K.gemm("fp16", a, w, bias=bias, out_f=False, out_h=True)
        check(lib().afx_debug_set(b"gemm_nodma", 8192 | bits))
        for _ in range(3):
            _, oh = K.gemm("fp16", a, w, bias=bias, out_f=False, out_h=True)
        torch.cuda.synchronize()
        check(lib().afx_debug_set(b"gemm_nodma", 0))
        ts = oh.view(torch.int64).flatten()[: 8 * 16 * 16].cpu().numpy().reshape(8, 16, 16).astype(np.int64)
        nk = min(Kk // 64, 16)
        print(f"== {name}: cycles (s_memtime), K-tiles 2..{nk - 2} of the first output tile of workgroup 0")
        for row, waves in (("row 0", range(0, 4)), ("row 1", range(4, 8))):
            seg = {k: [] for k in ("A read", "A wait1", "A mfma", "A wait2", "B read", "B wait1", "B mfma", "B wait2", "K-tile")}
            for wv in waves:
                for t in range(2, nk - 2):
                    s = ts[wv, t]
                    nxt = ts[wv, t + 1, 0]
                    for i, k in enumerate(("A read", "A wait1", "A mfma", "A wait2", "B read", "B wait1", "B mfma", "B wait2")):
                        seg[k].append(s[i + 1] - s[i])
                    seg["K-tile"].append(nxt - s[0])
            print(f"  {row}: " + "  ".join(f"{k} {np.mean(v):6.0f}" for k, v in seg.items()))
        # skew between the rows at the start of a K-tile
        print(f"  start of K-tile 4: wave 0 {ts[0, 4, 0] - ts[0, 4, 0]}, wave 4 {ts[4, 4, 0] - ts[0, 4, 0]} cycles after wave 0")
    check(lib().afx_debug_set(b"gemm_tile", -1))
    check(lib().afx_debug_set(b"gemm_fit", 1))


if __name__ == "__main__":
    main()
