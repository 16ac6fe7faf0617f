"""Where the split-precision Shaw attention differs from the fp64 definition (and from the fp32 VALU kernel): per N, the error by query
row, head dim, and a sweep that removes the relative term / flattens the softmax."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import kernels as K  # noqa: E402


def ref64(q, kv, rel, B, N, H, dh):
    qq = (q.double() * dh ** -0.5).view(B, N, H, dh).transpose(1, 2)
    kk = kv[:, : H * dh].double().reshape(B, N, H, dh).transpose(1, 2)
    vv = kv[:, H * dh:].double().reshape(B, N, H, dh).transpose(1, 2)
    seq = torch.arange(N)
    dist = (seq[:, None] - seq[None, :]).clamp(-512, 512) + 512
    dots = torch.einsum("bhid,bhjd->bhij", qq, kk)
    for bi in range(B):
        dots[bi] += torch.einsum("hnd,nrd->hnr", qq[bi], rel.double()[dist])
    return torch.einsum("bhij,bhjd->bhid", torch.softmax(dots, -1), vv).transpose(1, 2).reshape(B * N, H * dh), dots


B, H, dh = 2, 4, 36
for N in (13, 50, 200):
    for name, qs, rs in (("as in the test", 1.0, 1.0), ("no relative term", 1.0, 0.0), ("flat softmax (q x 0.1)", 0.1, 1.0), ("rel x 0.1", 1.0, 0.1)):
        g = torch.Generator().manual_seed(100 + N)
        q = torch.randn(B * N, H * dh, generator=g) * qs
        kv = torch.randn(B * N, 2 * H * dh, generator=g)
        rel = torch.randn(1025, dh, generator=g) * rs
        ref, dots = ref64(q, kv, rel, B, N, H, dh)
        got = K.conf_attn_mfma("fp16x3", q.cuda(), kv.cuda(), rel.cuda(), B, N, H, dh).cpu().double()
        valu = K.conf_attn("fp32", q.cuda(), kv.cuda(), rel.cuda(), B, N, H, dh).cpu().double()
        e, ev = (got - ref).abs(), (valu - ref).abs()
        row = int(e.max(dim=1)[0].argmax())
        print(f"N {N:3d} {name:24s}: split max {e.max():.2e} (row {row % N} of utt {row // N}, col {int(e[row].argmax())}), VALU max {ev.max():.2e}; max |logit| {dots.abs().max():.1f}; "
              f"rows with err > 1e-5: {(e.max(dim=1)[0] > 1e-5).sum().item()} of {B * N}", flush=True)

if os.environ.get("SHAW_DBG"):  # library built with -DSHAW_DBG: head dims 0 / 1 of every row hold the row's maximum logit and denominator
    for N in (50, 200):
        g = torch.Generator().manual_seed(100 + N)
        q = torch.randn(B * N, H * dh, generator=g)
        kv = torch.randn(B * N, 2 * H * dh, generator=g)
        rel = torch.randn(1025, dh, generator=g)
        ref, dots = ref64(q, kv, rel, B, N, H, dh)
        got = K.conf_attn_mfma("fp16x3", q.cuda(), kv.cuda(), rel.cuda(), B, N, H, dh).cpu().double().view(B, N, H, dh)
        mx_ref = dots.max(dim=-1)[0]                                  # (B, H, N)
        den_ref = torch.exp(dots - mx_ref[..., None]).sum(-1)
        mx_got, den_got = got[..., 0].transpose(1, 2), got[..., 1].transpose(1, 2)
        em, ed = (mx_got - mx_ref).abs(), ((den_got - den_ref) / den_ref).abs()
        i = int(ed.reshape(-1).argmax())
        print(f"N {N}: max |d max-logit| {em.max():.2e}; max relative |d denominator| {ed.max():.2e} at flat index {i} (b, h, row = {i // (H * N)}, {(i // N) % H}, {i % N}); "
              f"rows with relative denominator error > 1e-5: {(ed > 1e-5).sum().item()} of {B * H * N}", flush=True)
