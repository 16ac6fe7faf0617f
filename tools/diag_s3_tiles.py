"""Diagnostic (split precision): the tile instances the dispatcher can pick, forced one at a time for EVERY plain product of a
whole forward in dtype "fp16x3" (pair-form operands: twice the K-tiles of the fp16 product, 1.5 x the MFMAs per K-tile -- the
fill / bytes-per-FLOP trade-offs of the fp16 dispatch need not hold).  gemm_tile: -1 default dispatch, 0 = 128x128 2-stage,
3 = 8-phase 256-wide (fitted height), 5 = 128x64 2-stage, 8 = deep 128x64 (where the product is not narrow).  Per-class
times from the engine's profiler; logits must not move (same k order in every tile family)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import engine, synth  # noqa: E402
from afx._lib import lib  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "fp16x3"
for arch, oname, nl, B in (("xlsr_aasist", "XLSR_AASIST", 24, 16), ("conformer", "ConformerModel", 6, 64)):
    sd = synth.model_state_dict(oname, n_layers=nl)
    eng = engine.Engine(arch, n_layers=nl, dtype=dtype)
    eng.load_state_dict(sd)
    wave = synth.waveforms(B, 64000).cuda()
    ref = None
    for tile in (-1, 0, 3, 5, 8, -1):
        lib().afx_debug_set(b"gemm_tile", tile)
        out = eng.forward(wave).clone()
        ref = out if ref is None else ref
        for _ in range(2):
            eng.forward(wave)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            eng.forward(wave)
        e1.record()
        torch.cuda.synchronize()
        eng.profile_begin()
        for _ in range(4):
            eng.forward(wave)
        prof = eng.profile_end()
        brk = {k.replace("_kernel", ""): round(v["ms"] / 4, 3) for k, v in prof.items() if v["launches"] and k.startswith("gemm")}
        print(f"{arch} {dtype} B {B} gemm_tile {tile:2d}: {e0.elapsed_time(e1) / 8:.3f} ms / forward, bit-identical logits {torch.equal(out, ref)}  {brk}", flush=True)
    lib().afx_debug_set(b"gemm_tile", -1)
    del eng
    torch.cuda.empty_cache()
