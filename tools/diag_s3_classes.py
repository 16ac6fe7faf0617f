"""Per-class time (the engine's profiler, hipEvents on the launch stream) of one one-stream forward in a given precision, both models
at the bench batches.  usage: python tools/diag_s3_classes.py [dtype]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import engine, synth  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "fp16x3"
for arch, oname, nl, B in (("conformer", "ConformerModel", 6, 64), ("xlsr_aasist", "XLSR_AASIST", 24, 16)):
    eng = engine.Engine(arch, n_layers=nl, dtype=dtype)
    eng.load_state_dict(synth.model_state_dict(oname, n_layers=nl))
    wave = synth.waveforms(B, 64000).cuda()
    for _ in range(3):
        eng.forward(wave)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        eng.forward(wave)
    e1.record()
    torch.cuda.synchronize()
    eng.profile_begin()
    for _ in range(5):
        eng.forward(wave)
    prof = eng.profile_end()
    rows = sorted(((v["ms"] / 5, k, v["launches"] // 5) for k, v in prof.items() if v["launches"]), reverse=True)
    print(f"{arch} {dtype} B {B}: {e0.elapsed_time(e1) / 10:.3f} ms per one-stream forward; classes (ms, launches): " +
          ", ".join(f"{k.replace('_kernel', '')} {ms:.3f} ({n})" for ms, k, n in rows), flush=True)
    del eng
    torch.cuda.empty_cache()
