"""Split precision: the height of the 8-wave 256-wide tile (gemm_fit: 1 = the fp16-fitted cost model, 5..8 = forced fragments per wave row,
0 = always 256 rows) for whole forwards in dtype fp16x3 -- pair-form operands carry 3 MFMAs per fragment pair, the model rounds x (MF + 5)
was fitted on fp16.  Bit-identical logits expected (same k order at every height)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import engine, synth  # noqa: E402
from afx._lib import lib  # noqa: E402

for arch, oname, nl, B in (("xlsr_aasist", "XLSR_AASIST", 24, 16), ("conformer", "ConformerModel", 6, 64)):
    eng = engine.Engine(arch, n_layers=nl, dtype="fp16x3")
    eng.load_state_dict(synth.model_state_dict(oname, n_layers=nl))
    wave = synth.waveforms(B, 64000).cuda()
    ref = None
    for fit in (1, 5, 6, 7, 8, 0, 1):
        lib().afx_debug_set(b"gemm_fit", fit)
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
        print(f"{arch} fp16x3 B {B} gemm_fit {fit}: {e0.elapsed_time(e1) / 10:.3f} ms per one-stream forward, same logits {torch.equal(out, ref)}", flush=True)
    lib().afx_debug_set(b"gemm_fit", 1)
    del eng
    torch.cuda.empty_cache()
