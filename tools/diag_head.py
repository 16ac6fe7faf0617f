"""Diagnostic: the AASIST back-end alone (afx_head_forward) at a given batch of 199-frame windows -- the streaming scorers'
per-hop back-end.  Under rocprofv3 --kernel-trace --stats this lists where a window's 34 us go.
    python tools/diag_head.py [B] [iters]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import engine, synth  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    sd = synth.model_state_dict("XLSR_AASIST", n_layers=1)
    eng = engine.Engine("xlsr_aasist", n_layers=1, dtype="fp16")
    eng.load_state_dict(sd)
    feats = torch.randn(B, 199, 1024, device="cuda")
    for _ in range(2):
        eng.head(feats)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        eng.head(feats)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"AASIST back-end, B = {B}: {dt * 1e3:.3f} ms per call, {dt / B * 1e6:.1f} us per window", flush=True)

if __name__ == "__main__":
    main()
