"""Experiment: one batch of 64 as ONE forward against two half-batches on two streams (the HBM-bound kernels of
one half under the MFMA-bound kernels of the other).  Prints ms per 64 utterances for both."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import engine, synth  # noqa: E402

sd = synth.model_state_dict("ConformerModel", n_layers=6)
NS = int(os.environ.get("NS", "2"))
engs = [engine.Engine("conformer", n_layers=6, dtype="fp16") for _ in range(max(NS, 1))]
for e in engs:
    e.load_state_dict(sd)
wave = synth.waveforms(64, 64000).cuda()
sizes = [int(x) for x in os.environ.get("SPLIT", ",".join([str(64 // NS)] * NS)).split(",")]
assert sum(sizes) == 64 and len(sizes) == NS
offs = [sum(sizes[:i]) for i in range(NS + 1)]
halves = [wave[offs[i]:offs[i + 1]].contiguous() for i in range(NS)]
streams = [torch.cuda.Stream() for _ in range(NS)]


def one():
    return engs[0].forward(wave)


def two():
    outs = []
    cur = torch.cuda.current_stream()
    for e, h, s in zip(engs, halves, streams):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(e.forward(h))
    for s in streams:
        cur.wait_stream(s)
    return torch.cat(outs)


ref = one()
got = two()
print("max |diff| one vs two-stream:", (ref - got).abs().max().item())
for name, fn in (("one forward of 64", one), (f"{NS} streams {sizes}", two), ("one forward of 64", one), (f"{NS} streams {sizes}", two)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms", flush=True)
