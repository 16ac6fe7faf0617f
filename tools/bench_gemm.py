"""Micro-benchmark of the MFMA GEMM on the shapes the path launches (B=64 student
batch), for A/B-ing kernel variants in ONE process with interleaved rounds
(cdna guide rule 24).  Prints TFLOP/s per shape and variant (median of rounds)."""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import kernels as K  # noqa: E402
from afx._lib import lib, check  # noqa: E402

DT = os.environ.get("AFX_DTYPE", "fp16")
TD = torch.float16 if DT == "fp16" else torch.bfloat16
B = 64
VENDOR = None
# name, kind, dims
SHAPES = [
    ("conv1 M=409536 K=1536 N=512", "conv", (B, 12799, 3, 2)),
    ("conv3 M=102336 K=1536 N=512", "conv", (B, 3199, 3, 2)),
    ("conv5 M=25536  K=1024 N=512", "conv", (B, 799, 2, 2)),
    ("qkv   M=12736 K=1024 N=3072", "lin", (12736, 3072, 1024)),
    ("out   M=12736 K=1024 N=1024", "lin", (12736, 1024, 1024)),
    ("fc1   M=12736 K=1024 N=4096", "lin", (12736, 4096, 1024)),
    ("fc2   M=12736 K=4096 N=1024", "lin", (12736, 1024, 4096)),
    ("teacher fc1 M=3184 K=1024 N=4096", "lin", (3184, 4096, 1024)),
    ("teacher qkv M=3184 K=1024 N=3072", "lin", (3184, 3072, 1024)),
    ("teacher out M=3184 K=1024 N=1024", "lin", (3184, 1024, 1024)),
    ("teacher fc2 M=3184 K=4096 N=1024", "lin", (3184, 1024, 4096)),
    ("square 4096^3", "lin", (4096, 4096, 4096)),
    ("square 8192^3", "lin", (8192, 8192, 8192)),
]
SETS = {
    "tile": [("128x128", ("gemm_tile", 0)), ("256x256", ("gemm_tile", 1))],
    "8ph": [("128x128", ("gemm_tile", 0)), ("256x256 2-stage", ("gemm_tile", 1)), ("256x256 8-phase", ("gemm_tile", 3))],
    "w4": [("default dispatch", ("gemm_tile", -1)), ("8-phase 256-wide (fitted)", ("gemm_tile", 3)), ("4-wave 256x256 (AGPR acc)", ("gemm_tile", 4)),
           ("4-wave, 5-stage ring", ("gemm_tile", 10))],
    "small": [("default", ("gemm_tile", -1)), ("128x128", ("gemm_tile", 0)), ("128x64", ("gemm_tile", 5)), ("8-phase 256x256", ("gemm_tile", 3))],
    "split": [("single kernel", ("gemm_split", 0)), ("rounds + remainder", ("gemm_split", 1))],
    "map": [("map0", ("gemm_map", 0)), ("map1", ("gemm_map", 1)), ("map2", ("gemm_map", 2))],
    "nt": [("A default", ("gemm_a_nt", 0)), ("A nt", ("gemm_a_nt", 1))],
    "deep": [("2-stage", ("gemm_deep", 0)), ("8-phase", ("gemm_deep", 2))],
    # the vendor library on the same operands (torch.nn.functional.linear -> hipBLASLt / rocBLAS: bias epilogue only, no GELU, its own
    # choice of kernel per shape): a reference point for what the hardware gives these shapes, not a candidate for the path
    "vendor": [("this repository (bias + GELU epilogue)", ("gemm_tile", -1)), ("vendor library (bias only)", "vendor")],
    "epi": [("narrow stores", ("gemm_nodma", 16)), ("wide stores", ("gemm_nodma", 0))],  # attribution build only (AFX_LIB=.../libafx_attr.so)
}
VARIANTS = SETS[os.environ.get("BENCH_SET", "tile")]
if os.environ.get("BENCH_SET") in ("nt", "deep"):  # the fused conv + LayerNorm kernel (row-complete tile)
    SHAPES = [("convln1 M=409536 K=1536", "convln", (B, 12799, 3, 2)), ("convln2 M=204736 K=1536", "convln", (B, 6399, 3, 2)),
              ("convln3 M=102336 K=1536", "convln", (B, 3199, 3, 2)), ("convln5 M=25536 K=1024", "convln", (B, 799, 2, 2))] + SHAPES[3:]


def make(kind, dims):
    g = torch.Generator(device="cuda").manual_seed(1)
    if kind == "convln":
        Bb, Tin, k, s = dims
        x = torch.randn(Bb, Tin, 512, generator=g, device="cuda").to(TD)
        wp = (torch.randn(512, k * 512, generator=g, device="cuda") * 0.03).to(TD)
        bias = torch.randn(512, generator=g, device="cuda")
        ga = torch.ones(512, device="cuda")
        Tout = (Tin - k) // s + 1
        return (lambda: K.conv_ln_act(DT, x, wp, k, s, bias, ga, bias)), 2.0 * Bb * Tout * 512 * k * 512
    if kind == "conv":
        Bb, Tin, k, s = dims
        x = torch.randn(Bb, Tin, 512, generator=g, device="cuda").to(TD)
        wp = (torch.randn(512, k * 512, generator=g, device="cuda") * 0.03).to(TD)
        bias = torch.randn(512, generator=g, device="cuda")
        Tout = (Tin - k) // s + 1
        flops = 2.0 * Bb * Tout * 512 * k * 512
        return (lambda: K.conv_gemm(DT, x, wp, k, s, bias)), flops
    M, N, Kk = dims
    a = torch.randn(M, Kk, generator=g, device="cuda").to(TD)
    w = (torch.randn(N, Kk, generator=g, device="cuda") * 0.03).to(TD)
    bias = torch.randn(N, generator=g, device="cuda")
    global VENDOR
    bias_h = bias.to(TD)
    VENDOR = lambda: torch.nn.functional.linear(a, w, bias_h)
    return (lambda: K.gemm(DT, a, w, bias=bias, act="gelu", out_f=False, out_h=True)), 2.0 * M * N * Kk


def main():
    rounds, reps = 5, 10
    for name, kind, dims in SHAPES:
        if kind != "lin" and any(kv == "vendor" for _, kv in VARIANTS):
            continue
        fn, flops = make(kind, dims)
        mine = fn
        times = {v: [] for v, _ in VARIANTS}
        for _ in range(2):
            fn()
        for _ in range(rounds):
            for v, kv in VARIANTS:
                if kv == "vendor":
                    fn = VENDOR
                else:
                    fn = mine
                    for key, val in (kv if isinstance(kv, list) else [kv]):
                        check(lib().afx_debug_set(key.encode(), val))
                fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / reps)
        row = "  ".join(f"{v}: {flops / (statistics.median(t) * 1e-3) / 1e12:6.0f} TF ({statistics.median(t) * 1e3:6.1f} us)"
                        for v, t in times.items())
        print(f"{name:36s} {row}", flush=True)
    for key in ("gemm_tile", "gemm_map", "gemm_a_nt"):
        check(lib().afx_debug_set(key.encode(), -1))
    lib().afx_debug_set(b"gemm_nodma", 0)  # (refused by the product build: it has no such switch)
    check(lib().afx_debug_set(b"gemm_deep", -1))
    check(lib().afx_debug_set(b"gemm_split", 1))


if __name__ == "__main__":
    main()
