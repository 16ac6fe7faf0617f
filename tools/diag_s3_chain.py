"""fp16x3 Conformer head: fused row chains against the per-op path -- time of the student forward at the bench batch (one
stream, hipEvents around 20 forwards) and the profile classes of the head.  usage: python tools/diag_s3_chain.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "real-time-deepfake-speech-detection_amd"))
from afx import engine, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sd = synth.model_state_dict("ConformerModel", n_layers=6)
wave = synth.waveforms(B, 64000).cuda()
outs = {}
for fused in (1, 0, 1, 0):
    eng = engine.Engine("conformer", n_layers=6, dtype="fp16x3")
    eng.load_state_dict(sd)
    eng.set("fuse_conformer", fused)
    for _ in range(3):
        out = eng.forward(wave)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        out = eng.forward(wave)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    eng.profile_begin()
    for _ in range(5):
        eng.forward(wave)
    prof = eng.profile_end()
    head = {k: round(v["ms"] / 5, 3) for k, v in prof.items() if k.startswith("conf") and v["launches"]}
    outs[fused] = out.cpu()
    print(f"fuse_conformer={fused}: {ms:.3f} ms per forward of {B} ({B * 1e3 / ms:.0f} utt/s one-stream); head classes (ms): {head}")
print(f"max |dlogit| fused vs per-op: {(outs[1] - outs[0]).abs().max().item():.2e}")
