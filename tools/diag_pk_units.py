"""Round 4's two-stream defect, the other side of the build split: the translation units that KEEP packed fp32
math -- the GEMM tiles' epilogues, the attention kernels' softmax, the fused Conformer chains -- launched over and over on the main
stream beside the vendor library's GEMMs on a side stream (the neighbours that moved 45-57 of 60 launches of the packed VALU conv0
kernel, profiles/r04_conv0_pk_standalone.txt).  Every output must equal the first one bit for bit.
    python tools/diag_pk_units.py [launches per case]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import engine, kernels as K, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
g = torch.Generator(device="cuda").manual_seed(3)


def rnd(*shape, dt=torch.float32, scale=1.0):
    return (torch.randn(*shape, generator=g, device="cuda") * scale).to(dt)


# ---- victims: (name, callable -> tensor) ------------------------------------------------------------------------------
a_big, w_qkv, b_qkv = rnd(12736, 1024, dt=torch.float16), rnd(3072, 1024, dt=torch.float16, scale=0.03), rnd(3072)
w_fc1, b_fc1 = rnd(4096, 1024, dt=torch.float16, scale=0.03), rnd(4096)
a_t, w_out, b_out, resid = rnd(3184, 1024, dt=torch.float16), rnd(1024, 1024, dt=torch.float16, scale=0.03), rnd(1024), rnd(3184, 1024)
qkv = rnd(16 * 199, 3072, dt=torch.float16)
cq, ckv, crel = rnd(64 * 200, 144), rnd(64 * 200, 288), rnd(1025, 36, scale=0.1)
sd = synth.model_state_dict("ConformerModel", n_layers=1, n_encoders=2)
eng = engine.Engine("conformer", n_layers=1, dtype="fp16", conf_blocks=2)
eng.load_state_dict(sd)
feats = rnd(16, 199, 1024)
victims = [
    ("256-wide tile, bias (QKV 12736x3072x1024)", lambda: K.gemm("fp16", a_big, w_qkv, bias=b_qkv, out_f=False, out_h=True)[1]),
    ("256-wide tile, bias + GELU (FC1 12736x4096x1024)", lambda: K.gemm("fp16", a_big, w_fc1, bias=b_fc1, act="gelu", out_f=False, out_h=True)[1]),
    ("deep 128x64 tile, bias + residual, fp32 out (3184x1024x1024)", lambda: K.gemm("fp16", a_t, w_out, bias=b_out, resid=resid, out_f=True, out_h=False)[0]),
    ("transformer attention (B 16, T 199, 16 heads)", lambda: K.mhsa("fp16", qkv, 16, 199, 16)),
    ("Shaw attention on the matrix cores (B 64, N 200, 4 x 36)", lambda: K.conf_attn_mfma("fp16", cq, ckv, crel, 64, 200, 4, 36)),
    ("Conformer head: fused row chains + attention + depthwise conv (B 16)", lambda: eng.head(feats)),
]
# the split-precision forms of the same units (fp32 rows in / out)
qkv32 = rnd(16 * 199, 3072)
eng3 = engine.Engine("conformer", n_layers=1, dtype="fp16x3", conf_blocks=2)
eng3.load_state_dict(sd)
victims += [
    ("fp16x3: transformer attention (B 16, T 199)", lambda: K.mhsa("fp16x3", qkv32, 16, 199, 16)),
    ("fp16x3: Shaw attention on the matrix cores (B 64, N 200)", lambda: K.conf_attn_mfma("fp16x3", cq, ckv, crel, 64, 200, 4, 36)),
    ("fp16x3: Conformer head, fused split-precision chains (B 16)", lambda: eng3.head(feats)),
]
side = torch.cuda.Stream()


def mats(n, dt):
    return tuple(torch.randn(n, n, device="cuda").to(dt) for _ in range(2)) + (torch.empty(n, n, device="cuda", dtype=dt),)


aggressors = [("nothing", None, 0), ("vendor fp16 GEMM 2048^3 x 2", mats(2048, torch.float16), 2), ("vendor bf16 GEMM 2048^3 x 2", mats(2048, torch.bfloat16), 2),
              ("vendor fp16 GEMM 1024^3 x 6", mats(1024, torch.float16), 6), ("vendor fp16 GEMM 4096^3 x 1", mats(4096, torch.float16), 1)]
for an, m, reps in aggressors:
    if m is not None:
        torch.mm(m[0], m[1], out=m[2])
    for vn, fn in victims:
        ref = fn().clone()
        torch.cuda.synchronize()
        bad = 0
        for _ in range(N):
            if m is not None:
                with torch.cuda.stream(side):
                    for _ in range(reps):
                        torch.mm(m[0], m[1], out=m[2])
            out = fn()
            torch.cuda.synchronize()
            bad += not torch.equal(out, ref)
        print(f"{an:30s} | {vn:70s}: {bad} of {N} launches differ", flush=True)
