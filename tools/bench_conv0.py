"""A/B of conv layer 0's output stores (plain / non-temporal) inside the student forward: the layer's own time and the
time of conv layer 1 that reads its 843 MB right behind it (per-class hipEvent timing of the engine's profiler)."""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import engine, synth  # noqa: E402
from afx._lib import check, lib  # noqa: E402

sd = synth.model_state_dict("ConformerModel", n_layers=6)
eng = engine.Engine("conformer", n_layers=6, dtype="fp16")
eng.load_state_dict(sd)
wave = synth.waveforms(64, 64000).cuda()
res = {1: [], 3: []}
for _ in range(5):
    for mode in (1, 3):
        check(lib().afx_debug_set(b"conv0_mfma", mode))
        for _ in range(3):
            eng.forward(wave)
        torch.cuda.synchronize()
        eng.profile_begin()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            eng.forward(wave)
        e1.record()
        prof = eng.profile_end()
        torch.cuda.synchronize()
        res[mode].append((prof["conv0_kernel"]["ms"] / 10, prof["gemm8_kernel<128x512,rowLN>"]["ms"] / 10, e0.elapsed_time(e1) / 10))
check(lib().afx_debug_set(b"conv0_mfma", 1))
for mode, name in ((1, "plain stores"), (3, "non-temporal stores")):
    c0, cv, st = (statistics.median(x[i] for x in res[mode]) for i in range(3))
    print(f"{name:20s}: conv0 {c0 * 1e3:6.1f} us  conv layers 1-6 {cv * 1e3:7.1f} us  step {st:.3f} ms (instrumented)")
