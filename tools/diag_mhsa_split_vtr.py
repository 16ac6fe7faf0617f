"""Split-precision attention: row-major V + transposing reads (this build) against the library named by AFX_REF (the V^T scatter): same
bits, time per launch at the teacher's and the student's batch.  usage: AFX_LIB=<new> python tools/diag_mhsa_split_vtr.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import kernels as K  # noqa: E402

print("library:", os.environ.get("AFX_LIB", "product"))
g = torch.Generator(device="cuda").manual_seed(5)
for B, T in ((16, 199), (64, 199), (16, 49), (4, 120)):
    qkv = torch.randn(B * T, 3072, generator=g, device="cuda")
    out = K.mhsa("fp16x3", qkv, B, T, 16)
    ref = torch.nn.functional.scaled_dot_product_attention(*(qkv.double().reshape(B, T, 3, 16, 64).permute(2, 0, 3, 1, 4)), scale=0.125).permute(0, 2, 1, 3).reshape(B * T, 1024)
    err = (out.double() - ref).abs().max().item()
    for _ in range(5):
        K.mhsa("fp16x3", qkv, B, T, 16)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        K.mhsa("fp16x3", qkv, B, T, 16)
    e1.record()
    torch.cuda.synchronize()
    print(f"B {B:3d} T {T:3d}: max |d| vs fp64 attention {err:.2e}; {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us per launch; checksum {out.double().sum().item():.10e}", flush=True)
