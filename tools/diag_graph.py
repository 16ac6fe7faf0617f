"""Diagnostic: does replaying a forward as ONE hipGraph close the ~1 us gaps between its ~130-300 launches at the
benchmarked batch sizes?  (Engine.capture exists for B = 1 latency; this asks the throughput question.)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import engine, synth  # noqa: E402


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    for arch, oname, nl, B in (("conformer", "ConformerModel", 6, 64), ("xlsr_aasist", "XLSR_AASIST", 24, 16)):
        sd = synth.model_state_dict(oname, n_layers=nl)
        eng = engine.Engine(arch, n_layers=nl, dtype="fp16")
        eng.load_state_dict(sd)
        wave = synth.waveforms(B, 64000, batch_idx=0).cuda()
        plain = timeit(lambda: eng.forward(wave))
        run = eng.capture(B, 64000)
        assert torch.equal(run(wave), eng.forward(wave))
        graph = timeit(lambda: run(wave))
        two = timeit(lambda: eng.forward_overlapped(wave))
        eng.join()
        print(f"{arch} B={B}: one stream {plain:.3f} ms   hipGraph replay {graph:.3f} ms   two streams {two:.3f} ms", flush=True)
        del eng, run


if __name__ == "__main__":
    main()
