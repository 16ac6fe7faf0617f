"""A handful of GEMM launches on the path's shapes, for rocprofv3 --pmc passes
(one counter group per pass; see tools/pmc.sh)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import kernels as K  # noqa: E402

DT, TD, B = "fp16", torch.float16, 64
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(B, 12799, 512, generator=g, device="cuda").to(TD)
wp = (torch.randn(512, 1536, generator=g, device="cuda") * 0.03).to(TD)
bias = torch.randn(512, generator=g, device="cuda")
ga = torch.ones(512, device="cuda")
a = torch.randn(12736, 1024, generator=g, device="cuda").to(TD)
w1 = (torch.randn(4096, 1024, generator=g, device="cuda") * 0.03).to(TD)
b1 = torch.randn(4096, generator=g, device="cuda")
a4 = torch.randn(12736, 4096, generator=g, device="cuda").to(TD)
w2 = (torch.randn(1024, 4096, generator=g, device="cuda") * 0.03).to(TD)
b2 = torch.randn(1024, generator=g, device="cuda")
for _ in range(3):
    K.conv_ln_act(DT, x, wp, 3, 2, bias, ga, bias)            # row-complete 128x512 tile, conv1
    K.gemm(DT, a, w1, bias=b1, act="gelu", out_f=False, out_h=True)   # 128x128 tile, FC1
    K.gemm(DT, a4, w2, bias=b2, out_f=True, out_h=False)              # 256x256 tile, FC2
torch.cuda.synchronize()
