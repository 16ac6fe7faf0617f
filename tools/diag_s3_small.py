"""Diagnostic (split precision): the tile of the teacher's small-M products (out-proj, FC2: M = 16 x 199, N = 1024) -- the fp16
dispatch's deep 128x64 tile against the 128x128 and the 2-stage 128x64 tiles, everything else on the default dispatch.
Pair-form operands carry 3 MFMAs per fragment pair: the bytes-per-FLOP balance that chose the tile for fp16 need not hold."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import engine, synth  # noqa: E402
from afx._lib import lib  # noqa: E402

for B in (16, 8):
    sd = synth.model_state_dict("XLSR_AASIST", n_layers=24)
    eng = engine.Engine("xlsr_aasist", n_layers=24, dtype="fp16x3")
    eng.load_state_dict(sd)
    wave = synth.waveforms(B, 64000).cuda()
    ref = None
    for knob in (0, 1, 2, 0, 1, 2):
        lib().afx_debug_set(b"gemm_s3_small", knob)
        out = eng.forward(wave).clone()
        ref = out if ref is None else ref
        for _ in range(2):
            eng.forward(wave)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            eng.forward(wave)
        e1.record()
        torch.cuda.synchronize()
        eng.profile_begin()
        for _ in range(4):
            eng.forward(wave)
        prof = eng.profile_end()
        brk = {k.replace("_kernel", ""): round(v["ms"] / 4, 3) for k, v in prof.items() if v["launches"] and k.startswith("gemm")}
        print(f"teacher fp16x3 B {B} gemm_s3_small {knob}: {e0.elapsed_time(e1) / 10:.3f} ms / forward, bit-identical logits {torch.equal(out, ref)}  {brk}", flush=True)
    lib().afx_debug_set(b"gemm_s3_small", 0)
    del eng
    torch.cuda.empty_cache()
