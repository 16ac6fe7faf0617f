"""Python faces of the single-kernel C entry points (afx_k_*), used by the unit
parity tests.  All tensors must be contiguous CUDA tensors; ``dtype`` is "bf16" or
"fp16" and names the matrix-core operand type of the half-precision arguments."""
import torch

from ._lib import ACT_GELU, ACT_NONE, ACT_SELU, ACT_SWISH, call_on, check, lib, ptr, stream_ptr
from .engine import DTYPES, torch_dtype

ACTS = {None: ACT_NONE, "gelu": ACT_GELU, "swish": ACT_SWISH, "selu": ACT_SELU}


def gemm(dtype, A, W, bias=None, act=None, alpha=1.0, resid=None, out_f=True, out_h=False):
    """A (M,K) half, W (N,K) half -> fp32 and/or half (M,N).  dtype "fp16x3" (split precision): A and W are FP32 and the
    "operand type" output is fp32 too; the entry point builds the hi / lo operand forms per call (test hook)."""
    M, K = A.shape
    N = W.shape[0]
    of = torch.empty(M, N, dtype=torch.float32, device=A.device) if out_f else None
    oh = torch.empty(M, N, dtype=torch_dtype(dtype), device=A.device) if out_h else None
    check(call_on(A, lib().afx_k_gemm, DTYPES[dtype], ptr(A), A.stride(0), ptr(W), W.stride(0), M, N, K, ptr(bias), ACTS[act],
                           alpha, ptr(resid), N, ptr(of), N, ptr(oh), N))
    return of, oh


def pack_linear(dtype, w, kpad=None):
    N, K = w.shape
    kpad = kpad or K
    out = torch.empty(N, kpad, dtype=torch_dtype(dtype), device=w.device)
    check(call_on(w, lib().afx_k_pack_linear, DTYPES[dtype], ptr(w), N, K, kpad, ptr(out)))
    return out


def pack_conv(dtype, w):
    N, Cin, k = w.shape
    out = torch.empty(N, k * Cin, dtype=torch_dtype(dtype), device=w.device)
    check(call_on(w, lib().afx_k_pack_conv, DTYPES[dtype], ptr(w), N, Cin, k, ptr(out)))
    return out


def conv_gemm(dtype, x_h, wp, k, s, bias=None):
    """x_h (B,Tin,Cin) half channel-last, wp (N,k*Cin) packed -> (B,Tout,N) fp32."""
    B, Tin, Cin = x_h.shape
    N = wp.shape[0]
    Tout = (Tin - k) // s + 1
    out = torch.empty(B, Tout, N, dtype=torch.float32, device=x_h.device)
    check(call_on(x_h, lib().afx_k_conv_gemm, DTYPES[dtype], ptr(x_h), ptr(wp), B, Tin, Tout, Cin, k, s, N, ptr(bias), ptr(out)))
    return out


def conv_ln_act(dtype, x_h, wp, k, s, bias, gamma, beta, act="gelu", eps=1e-5, out_f=False, out_h=True):
    """Conv1d(512->512,k,s) + LayerNorm(512) + activation in one kernel; x_h (B,Tin,512) half."""
    B, Tin, Cin = x_h.shape
    Tout = (Tin - k) // s + 1
    of = torch.empty(B, Tout, 512, dtype=torch.float32, device=x_h.device) if out_f else None
    oh = torch.empty(B, Tout, 512, dtype=torch_dtype(dtype), device=x_h.device) if out_h else None
    check(call_on(x_h, lib().afx_k_conv_ln_act, DTYPES[dtype], ptr(x_h), ptr(wp), B, Tin, Tout, Cin, k, s, ptr(bias), ptr(gamma),
                                  ptr(beta), eps, ACTS[act], ptr(of), ptr(oh)))
    return of, oh


def conv0(dtype, wave, w, bias, gamma, beta, pre_emph=False, coef=0.97):
    B, L = wave.shape
    T0 = (L - 10) // 5 + 1
    out = torch.empty(B, T0, 512, dtype=torch_dtype(dtype), device=wave.device)
    check(call_on(wave, lib().afx_k_conv0, DTYPES[dtype], ptr(wave), B, L, ptr(w), ptr(bias), ptr(gamma), ptr(beta),
                            1 if pre_emph else 0, coef, ptr(out)))
    return out


def conv0_pack(w, bias):
    """The layer's split-precision fp16 operand block, built once per checkpoint (afx_k_conv0_pack)."""
    pack = torch.empty(lib().afx_k_conv0_pack_bytes(), dtype=torch.uint8, device=w.device)
    check(call_on(w, lib().afx_k_conv0_pack, ptr(w), ptr(bias), ptr(pack)))
    return pack


def conv0_packed(dtype, wave, pack, w, bias, gamma, beta, pre_emph=False, coef=0.97):
    """conv0 with the operand block passed in: asynchronous, no allocation inside the library (the streaming hot path)."""
    B, L = wave.shape
    T0 = (L - 10) // 5 + 1
    out = torch.empty(B, T0, 512, dtype=torch_dtype(dtype), device=wave.device)
    check(call_on(wave, lib().afx_k_conv0_packed, DTYPES[dtype], ptr(wave), B, L, ptr(pack), ptr(w), ptr(bias), ptr(gamma),
                  ptr(beta), 1 if pre_emph else 0, coef, ptr(out)))
    return out


def rownorm(dtype, x, gamma, beta, eps=1e-5, act=None, out_f=True, out_h=False):
    rows, Cc = x.shape
    of = torch.empty(rows, Cc, dtype=torch.float32, device=x.device) if out_f else None
    oh = torch.empty(rows, Cc, dtype=torch_dtype(dtype), device=x.device) if out_h else None
    check(call_on(x, lib().afx_k_rownorm, DTYPES[dtype], ptr(x), x.stride(0), rows, Cc, ptr(gamma), ptr(beta), eps, ACTS[act],
                              ptr(of), Cc, ptr(oh), Cc))
    return of, oh


def mhsa(dtype, qkv, B, T, H):
    """qkv (B*T, 3*H*64) half -> (B*T, H*64) half."""
    out = torch.empty(B * T, H * 64, dtype=torch_dtype(dtype), device=qkv.device)
    check(call_on(qkv, lib().afx_k_mhsa, DTYPES[dtype], ptr(qkv), ptr(out), B, T, H))
    return out


def conf_attn(dtype, q, kv, rel, B, N, H, dh, max_pos=512):
    """q (B*N,H*dh) fp32, kv (B*N,2*H*dh) fp32, rel (2*max_pos+1,dh) -> (B*N,H*dh) half."""
    out = torch.empty(B * N, H * dh, dtype=torch_dtype(dtype), device=q.device)
    check(call_on(q, lib().afx_k_conf_attn, DTYPES[dtype], ptr(q), q.stride(0), ptr(kv), kv.stride(0), ptr(rel), max_pos, B, N, H,
                                dh, ptr(out), H * dh))
    return out


def conf_attn_mfma(dtype, q, kv, rel, B, N, H, dh, max_pos=512):
    """Matrix-core form of conf_attn (head dim 36, N <= 209): the table is packed to (2*max_pos+1, 64) halfs first."""
    rel_h = pack_linear("fp32" if dtype == "fp16x3" else dtype, rel, 64)  # (split precision: fp32 rows, split inside the kernel)
    out = torch.empty(B * N, H * dh, dtype=torch_dtype(dtype), device=q.device)
    check(call_on(q, lib().afx_k_conf_attn_mfma, DTYPES[dtype], ptr(q), q.stride(0), ptr(kv), kv.stride(0), ptr(rel_h), max_pos, B, N,
                                     H, dh, ptr(out), H * dh))
    return out


def conf_dwconv(dtype, x, w, bias, bn_scale, bn_shift, B, N, Cc, k):
    """x (B*N, 2*C) fp32 -> (B*N, C) half."""
    out = torch.empty(B * N, Cc, dtype=torch_dtype(dtype), device=x.device)
    check(call_on(x, lib().afx_k_conf_dwconv, DTYPES[dtype], ptr(x), x.stride(0), ptr(w), ptr(bias), ptr(bn_scale), ptr(bn_shift),
                                  B, N, Cc, k, ptr(out), Cc))
    return out


# ---- AASIST graph modules (fp32) ----------------------------------------------------
def _check_aasist(rc):
    if rc != 0:
        from ._lib import AfxError
        raise AfxError(lib().afx_aasist_error().decode())


def bn_fold(weight, bias, mean, var, eps=1e-5):
    """Eval BatchNorm as (scale, shift).  Parameter preparation, not hot-path compute."""
    scale = weight / torch.sqrt(var + eps)
    return scale.contiguous(), (bias - mean * scale).contiguous()


def gat(x, p, temp):
    """GraphAttentionLayer (models/aasist_modules.py:17-110).  p: dict of CUDA fp32
    tensors att_w, att_b, att_vec, w1, b1, w2, b2, bn_scale, bn_shift."""
    B, N, din = x.shape
    dout = p["att_w"].shape[0]
    y = torch.empty(B, N, dout, dtype=torch.float32, device=x.device)
    _check_aasist(call_on(x, lib().afx_k_gat, ptr(x), B, N, din, dout, ptr(p["att_w"]), ptr(p["att_b"]), ptr(p["att_vec"]),
                                  ptr(p["w1"]), ptr(p["b1"]), ptr(p["w2"]), ptr(p["b2"]), ptr(p["bn_scale"]),
                                  ptr(p["bn_shift"]), temp, ptr(y)))
    return y


def resblock(x, conv1_w, conv1_b, bn2_scale, bn2_shift, conv2_w, conv2_b, down_w=None, down_b=None):
    """Residual_block (models/aasist_modules.py:340-397) on an NCHW fp32 image; conv weights in checkpoint layout."""
    B, cin, H, W = x.shape
    cout = conv1_w.shape[0]
    scratch = torch.empty(lib().afx_k_resblock_scratch_floats(B, cin, cout, H, W), dtype=torch.float32, device=x.device)
    y = torch.empty(B, cout, H, W, dtype=torch.float32, device=x.device)
    _check_aasist(call_on(x, lib().afx_k_resblock, ptr(x), B, cin, cout, H, W, ptr(conv1_w), ptr(conv1_b), ptr(bn2_scale),
                          ptr(bn2_shift), ptr(conv2_w), ptr(conv2_b), ptr(down_w), ptr(down_b), ptr(scratch), ptr(y)))
    return y


HGAT_ORDER = ["t1w", "t1b", "t2w", "t2b", "att_w", "att_b", "attM_w", "attM_b", "v11", "v22", "v12", "vM",
              "w1", "b1", "w2", "b2", "w1M", "b1M", "w2M", "b2M", "bn_scale", "bn_shift"]


def hgat(x1, x2, p, temp, master=None):
    """HtrgGraphAttentionLayer (models/aasist_modules.py:112-294)."""
    import ctypes as C
    B, n1, din = x1.shape
    n2 = x2.shape[1]
    dout = p["att_w"].shape[0]
    dev = x1.device
    y1 = torch.empty(B, n1, dout, dtype=torch.float32, device=dev)
    y2 = torch.empty(B, n2, dout, dtype=torch.float32, device=dev)
    mo = torch.empty(B, 1, dout, dtype=torch.float32, device=dev)
    scratch = torch.empty(B * (n1 + n2) * din + B * din, dtype=torch.float32, device=dev)
    arr = (C.c_void_p * len(HGAT_ORDER))(*[p[k].data_ptr() for k in HGAT_ORDER])
    mstride = 0
    if master is not None:
        master = master.contiguous()
        mstride = 0 if master.shape[0] == 1 else din  # a (1,1,D) parameter is shared by the batch
    _check_aasist(call_on(x1, lib().afx_k_hgat, ptr(x1), n1, ptr(x2), n2, B, din, dout, arr, temp, ptr(master), mstride,
                                   ptr(scratch), ptr(y1), ptr(y2), ptr(mo)))
    return y1, y2, mo


def graph_pool(h, w, b, k):
    """GraphPool (models/aasist_modules.py:296-338)."""
    B, N, D = h.shape
    keep = max(int(N * k), 1)
    out = torch.empty(B, keep, D, dtype=torch.float32, device=h.device)
    _check_aasist(call_on(h, lib().afx_k_graph_pool, ptr(h), B, N, D, keep, ptr(w), ptr(b), ptr(out)))
    return out
