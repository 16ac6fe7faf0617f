"""Diagnostic (round 4's two-stream defect): conv layer 0's VALU kernel ALONE as the victim -- no engine, no
trunk -- launched over and over on the main stream while the vendor library's GEMMs run on a side stream.  Run once with the product
library (tap loops scalar) and once with the diagnostic variant whose loops are packed as hipcc packs them:

    make -C real-time-deepfake-speech-detection_amd/csrc variant NAME=c0pk DEFS=-DAFX_C0_PACKED
    AFX_LIB=$PWD/real-time-deepfake-speech-detection_amd/lib/libafx_c0pk.so python tools/diag_conv0_pk.py

Aggressors: none; fp16 GEMMs of three sizes (many short launches ... one long launch per victim launch); an fp32 GEMM long enough to
overlap; bf16.  Every output must equal the first one bit for bit."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import kernels as K, synth  # noqa: E402
from afx._lib import lib  # noqa: E402

print("library:", os.environ.get("AFX_LIB", "product"), lib().afx_build_id().decode() if hasattr(lib(), "afx_build_id") else "")
sd = synth.model_state_dict("XLSR_AASIST", n_layers=1)
P = "ssl_model.model.feature_extractor.conv_layers.0."
w, b, g, be = (sd[P + k].cuda() for k in ("0.weight", "0.bias", "2.1.weight", "2.1.bias"))
wave = synth.waveforms(16, 64000, batch_idx=701).cuda()  # 16 x 12 799 frames: ~100 us of the VALU kernel
ref = K.conv0("fp32", wave, w.reshape(512, 10), b, g, be).clone()
side = torch.cuda.Stream()


def mats(n, dt):
    return tuple(torch.randn(n, n, device="cuda").to(dt) for _ in range(2)) + (torch.empty(n, n, device="cuda", dtype=dt),)


cases = [("nothing beside", None, 0), ("fp16 GEMM 512^3 x 16 (short launches)", mats(512, torch.float16), 16),
         ("fp16 GEMM 2048^3 x 2", mats(2048, torch.float16), 2), ("fp16 GEMM 8192^3 x 1 (one long launch)", mats(8192, torch.float16), 1),
         ("bf16 GEMM 2048^3 x 2", mats(2048, torch.bfloat16), 2), ("fp32 GEMM 2048^3 x 1", mats(2048, torch.float32), 1)]
# this library's own GEMM kernels as aggressors: the 8-wave 256-wide tile (160 KB of LDS: nothing shares its CU) and a small tile
ga, gw = torch.randn(4096, 1024, device="cuda").half(), (torch.randn(4096, 1024, device="cuda") * 0.03).half()
sa, sw = torch.randn(2048, 1024, device="cuda").half(), (torch.randn(256, 1024, device="cuda") * 0.03).half()
cases += [("afx 256-wide tile 4096x4096x1024 x 2", ("afx", ga, gw), 2), ("afx small tile 2048x256x1024 x 8", ("afx", sa, sw), 8)]


def beside(m):
    if m[0] == "afx":
        K.gemm("fp16", m[1], m[2], out_f=False, out_h=True)
    else:
        torch.mm(m[0], m[1], out=m[2])


for name, m, reps in cases:
    if m is not None:
        beside(m)
    torch.cuda.synchronize()
    bad, rows_bad, lanes = 0, 0, set()
    for it in range(60):
        if m is not None:
            with torch.cuda.stream(side):
                for _ in range(reps):
                    beside(m)
        out = K.conv0("fp32", wave, w.reshape(512, 10), b, g, be)
        torch.cuda.synchronize()
        if not torch.equal(out, ref):
            bad += 1
            d = (out != ref).reshape(-1, 512)
            rows = d.any(dim=1).nonzero().flatten()
            rows_bad += rows.numel()
            big = ((out - ref).abs().reshape(-1, 512)[rows[0]] > 1e-3).nonzero().flatten().tolist()
            lanes.update((c // 8, c % 8) for c in big)
    ls = sorted(lanes)
    print(f"{name:42s}: {bad:2d} of 60 launches differ ({rows_bad} frames); (lane, element) of the large errors: "
          f"lanes {sorted({l for l, _ in ls})[:3]}..{sorted({l for l, _ in ls})[-3:] if ls else []} elements {sorted({e for _, e in ls})}", flush=True)
