"""Single-utterance latency of the student path, eager launches vs hipGraph replay."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import engine, synth
sd = synth.model_state_dict("ConformerModel", n_layers=6)
eng = engine.Engine("conformer", n_layers=6, dtype="fp16")
eng.load_state_dict(sd)
for B, L in ((1, 64000), (1, 16000), (8, 64000)):
    wave = synth.waveforms(B, L).cuda()
    run = eng.capture(B, L)
    for name, fn in (("eager", lambda: eng.forward(wave)), ("hipGraph", lambda: run(wave))):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 200
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"B={B} L={L} {name:8s} {dt * 1e3:7.3f} ms per forward  ({B / dt:8.1f} utt/s)")
