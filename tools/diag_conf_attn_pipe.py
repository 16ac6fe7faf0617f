"""Shaw attention on the matrix cores (conf_attn_mfma_kernel): time per launch and a checksum, for A/B between two builds of the
library (AFX_LIB).  usage: [AFX_LIB=...] python tools/diag_conf_attn_pipe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import kernels as K  # noqa: E402

print("library:", os.environ.get("AFX_LIB", "product"))
g = torch.Generator(device="cuda").manual_seed(7)
for B, N in ((64, 200), (16, 200), (8, 120)):
    q = torch.randn(B * N, 144, generator=g, device="cuda")
    kv = torch.randn(B * N, 288, generator=g, device="cuda")
    rel = torch.randn(1025, 36, generator=g, device="cuda") * 0.1
    out = K.conf_attn_mfma("fp16", q, kv, rel, B, N, 4, 36)
    from afx.kernels import pack_linear, lib, ptr, check, call_on, DTYPES  # the table packed once: time the kernel, not the pack
    rel_h = pack_linear("fp16", rel, 64)
    o = torch.empty(B * N, 144, dtype=torch.float16, device="cuda")

    def run():
        check(call_on(q, lib().afx_k_conf_attn_mfma, DTYPES["fp16"], ptr(q), q.stride(0), ptr(kv), kv.stride(0), ptr(rel_h), 512, B, N, 4, 36, ptr(o), 144))
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(f"B {B:3d} N {N:3d}: {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us per launch; checksum {out.double().sum().item():.10e} / {o.double().sum().item():.10e}", flush=True)
    # split precision: the fp32 VALU kernel against the hi / lo MFMA kernel (dtype "fp16x3")
    rel32 = pack_linear("fp32", rel, 64)
    o32 = torch.empty(B * N, 144, dtype=torch.float32, device="cuda")

    def run_valu():
        check(call_on(q, lib().afx_k_conf_attn, DTYPES["fp32"], ptr(q), q.stride(0), ptr(kv), kv.stride(0), ptr(rel), 512, B, N, 4, 36, ptr(o32), 144))

    def run_split():
        check(call_on(q, lib().afx_k_conf_attn_mfma, DTYPES["fp16x3"], ptr(q), q.stride(0), ptr(kv), kv.stride(0), ptr(rel32), 512, B, N, 4, 36, ptr(o32), 144))
    for name, fn in (("fp32 VALU kernel", run_valu), ("split-precision MFMA kernel", run_split)):
        try:
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(30):
                fn()
            e1.record()
            torch.cuda.synchronize()
            print(f"      {name:28s}: {e0.elapsed_time(e1) / 30 * 1e3:7.1f} us per launch; checksum {o32.double().sum().item():.10e}", flush=True)
        except Exception as exc:
            print(f"      {name}: {str(exc)[:100]}")
