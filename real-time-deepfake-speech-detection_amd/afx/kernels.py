"""Python faces of the single-kernel C entry points (afx_k_*), used by the unit
parity tests.  All tensors must be contiguous CUDA tensors; ``dtype`` is "bf16" or
"fp16" and names the matrix-core operand type of the half-precision arguments."""
import torch

from ._lib import ACT_GELU, ACT_NONE, ACT_SELU, ACT_SWISH, check, lib, ptr, stream_ptr
from .engine import DTYPES, torch_dtype

ACTS = {None: ACT_NONE, "gelu": ACT_GELU, "swish": ACT_SWISH, "selu": ACT_SELU}


def gemm(dtype, A, W, bias=None, act=None, alpha=1.0, resid=None, out_f=True, out_h=False):
    """A (M,K) half, W (N,K) half -> fp32 and/or half (M,N)."""
    M, K = A.shape
    N = W.shape[0]
    of = torch.empty(M, N, dtype=torch.float32, device=A.device) if out_f else None
    oh = torch.empty(M, N, dtype=torch_dtype(dtype), device=A.device) if out_h else None
    check(lib().afx_k_gemm(DTYPES[dtype], ptr(A), A.stride(0), ptr(W), W.stride(0), M, N, K, ptr(bias), ACTS[act],
                           alpha, ptr(resid), N, ptr(of), N, ptr(oh), N, stream_ptr()))
    return of, oh


def pack_linear(dtype, w, kpad=None):
    N, K = w.shape
    kpad = kpad or K
    out = torch.empty(N, kpad, dtype=torch_dtype(dtype), device=w.device)
    check(lib().afx_k_pack_linear(DTYPES[dtype], ptr(w), N, K, kpad, ptr(out), stream_ptr()))
    return out


def pack_conv(dtype, w):
    N, Cin, k = w.shape
    out = torch.empty(N, k * Cin, dtype=torch_dtype(dtype), device=w.device)
    check(lib().afx_k_pack_conv(DTYPES[dtype], ptr(w), N, Cin, k, ptr(out), stream_ptr()))
    return out


def conv_gemm(dtype, x_h, wp, k, s, bias=None):
    """x_h (B,Tin,Cin) half channel-last, wp (N,k*Cin) packed -> (B,Tout,N) fp32."""
    B, Tin, Cin = x_h.shape
    N = wp.shape[0]
    Tout = (Tin - k) // s + 1
    out = torch.empty(B, Tout, N, dtype=torch.float32, device=x_h.device)
    check(lib().afx_k_conv_gemm(DTYPES[dtype], ptr(x_h), ptr(wp), B, Tin, Tout, Cin, k, s, N, ptr(bias), ptr(out),
                                stream_ptr()))
    return out


def conv0(dtype, wave, w, bias, gamma, beta, pre_emph=False, coef=0.97):
    B, L = wave.shape
    T0 = (L - 10) // 5 + 1
    out = torch.empty(B, T0, 512, dtype=torch_dtype(dtype), device=wave.device)
    check(lib().afx_k_conv0(DTYPES[dtype], ptr(wave), B, L, ptr(w), ptr(bias), ptr(gamma), ptr(beta),
                            1 if pre_emph else 0, coef, ptr(out), stream_ptr()))
    return out


def rownorm(dtype, x, gamma, beta, eps=1e-5, act=None, out_f=True, out_h=False):
    rows, Cc = x.shape
    of = torch.empty(rows, Cc, dtype=torch.float32, device=x.device) if out_f else None
    oh = torch.empty(rows, Cc, dtype=torch_dtype(dtype), device=x.device) if out_h else None
    check(lib().afx_k_rownorm(DTYPES[dtype], ptr(x), x.stride(0), rows, Cc, ptr(gamma), ptr(beta), eps, ACTS[act],
                              ptr(of), Cc, ptr(oh), Cc, stream_ptr()))
    return of, oh


def mhsa(dtype, qkv, B, T, H):
    """qkv (B*T, 3*H*64) half -> (B*T, H*64) half."""
    out = torch.empty(B * T, H * 64, dtype=torch_dtype(dtype), device=qkv.device)
    check(lib().afx_k_mhsa(DTYPES[dtype], ptr(qkv), ptr(out), B, T, H, stream_ptr()))
    return out


def conf_attn(dtype, q, kv, rel, B, N, H, dh, max_pos=512):
    """q (B*N,H*dh) fp32, kv (B*N,2*H*dh) fp32, rel (2*max_pos+1,dh) -> (B*N,H*dh) half."""
    out = torch.empty(B * N, H * dh, dtype=torch_dtype(dtype), device=q.device)
    check(lib().afx_k_conf_attn(DTYPES[dtype], ptr(q), q.stride(0), ptr(kv), kv.stride(0), ptr(rel), max_pos, B, N, H,
                                dh, ptr(out), H * dh, stream_ptr()))
    return out


def conf_dwconv(dtype, x, w, bias, bn_scale, bn_shift, B, N, Cc, k):
    """x (B*N, 2*C) fp32 -> (B*N, C) half."""
    out = torch.empty(B * N, Cc, dtype=torch_dtype(dtype), device=x.device)
    check(lib().afx_k_conf_dwconv(DTYPES[dtype], ptr(x), x.stride(0), ptr(w), ptr(bias), ptr(bn_scale), ptr(bn_shift),
                                  B, N, Cc, k, ptr(out), Cc, stream_ptr()))
    return out
