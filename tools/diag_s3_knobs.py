"""Diagnostic (split precision, dtype "fp16x3"): the conv stack with the LayerNorm + GELU in the product's epilogue (the
row-complete 128x512 tile, pair-form walk) against the two-kernel form (256-wide tile + a LayerNorm pass that writes the next
layer's pair-form operand), per engine knob "fuse_conv_ln"; both workloads, one-stream forwards."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import engine, synth  # noqa: E402


def ms(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for arch, oname, nl, B in (("conformer", "ConformerModel", 6, 64), ("xlsr_aasist", "XLSR_AASIST", 24, 16)):
    sd = synth.model_state_dict(oname, n_layers=nl)
    eng = engine.Engine(arch, n_layers=nl, dtype="fp16x3")
    eng.load_state_dict(sd)
    wave = synth.waveforms(B, 64000).cuda()
    ref = None
    for rep in range(2):
        for fuse in (1, 0):
            eng.set("fuse_conv_ln", fuse)
            out = eng.forward(wave).clone()
            ref = out if ref is None else ref
            t = ms(lambda: eng.forward(wave))
            eng.profile_begin()
            for _ in range(5):
                eng.forward(wave)
            prof = eng.profile_end()
            brk = {k: round(v["ms"] / 5, 3) for k, v in prof.items() if v["launches"]}
            print(f"{arch} B {B} fuse_conv_ln {fuse}: {t:.3f} ms / forward, max |dlogit| vs fused {float((out - ref).abs().max()):.2e}  {brk}", flush=True)
    del eng
    torch.cuda.empty_cache()
