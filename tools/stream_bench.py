"""Real-time factor of the sliding-window streaming mode (BASELINE config 5) on one GPU:
S concurrent streams, 250 ms hops, every hop re-scores each stream's last 4 s with the student."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import engine, synth  # noqa: E402
from afx.streaming import SlidingWindowScorer  # noqa: E402

sd = synth.model_state_dict("ConformerModel", n_layers=6)
eng = engine.Engine("conformer", n_layers=6, dtype="fp16")
eng.load_state_dict(sd)
W, H = 64000, 4000
for S in [int(a) for a in (sys.argv[1:] or ["1", "64", "512", "2048"])]:
    sc = SlidingWindowScorer(eng, S, window=W, hop=H)
    chunk = (0.1 * torch.randn(S, H)).cuda()
    for _ in range(W // H + 2):  # fill the rings, reach the steady state
        sc.push(chunk)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        scores = sc.push(chunk)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"streams {S:5d}: {dt * 1e3:8.2f} ms per 250-ms hop  RTF {dt / 0.25:6.3f}  ({S / dt:8.0f} window scores/s)", flush=True)
    del sc
