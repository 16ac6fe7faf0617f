"""Diagnostic: one 4-s window (B = 1) -- eager launches against one hipGraph replay (Engine.capture), both heads."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import engine, synth  # noqa: E402


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
        torch.cuda.synchronize()  # a real-time caller reads every score
    return (time.perf_counter() - t0) / n * 1e3


def main():
    for arch, oname, nl in (("conformer", "ConformerModel", 6), ("xlsr_aasist", "XLSR_AASIST", 24)):
        sd = synth.model_state_dict(oname, n_layers=nl)
        eng = engine.Engine(arch, n_layers=nl, dtype="fp16")
        eng.load_state_dict(sd)
        for B in (1, 4):
            wave = synth.waveforms(B, 64000, batch_idx=0).cuda()
            eager = timeit(lambda: eng.forward(wave))
            run = eng.capture(B, 64000)
            assert torch.equal(run(wave), eng.forward(wave))
            graph = timeit(lambda: run(wave))
            print(f"{arch} B={B}: eager {eager:.3f} ms   hipGraph replay {graph:.3f} ms (score read every call)", flush=True)
        del eng


if __name__ == "__main__":
    main()
