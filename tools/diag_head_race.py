"""Diagnostic: is the AASIST back-end's result stable when ANOTHER engine's trunk runs beside it on a second stream?  (A kernel
with a latent intra-workgroup race -- a missing barrier, an LDS buffer reused too early -- gives the same bits as long as its
waves are scheduled the same way, and other bits under contention.)  Taps of the back-end's stages say where it first moves."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import engine, synth  # noqa: E402

sd = synth.model_state_dict("XLSR_AASIST", n_layers=2, head_scale=1.5)
for hd, td in (("fp16x3", "fp16x3"), ("fp16", "fp16x3"), ("fp16x3", "fp16"), ("fp16", "fp16")):
    head_eng = engine.Engine("xlsr_aasist", n_layers=2, dtype=hd)
    head_eng.load_state_dict(sd)
    trunk_eng = engine.Engine("xlsr_aasist", n_layers=2, dtype=td)
    trunk_eng.load_state_dict(sd)
    waves = [synth.waveforms(b, 16000, batch_idx=700 + i).cuda() for i, b in enumerate([5, 5, 5, 3, 7, 5, 1])]
    feats = [head_eng.ssl(w).clone() for w in waves]
    names = ["e_S", "e_T", "hidden"]
    head_eng.enable_taps()
    alone = []
    for f in feats:
        out = head_eng.head(f).clone()
        alone.append((out, [head_eng.tap(n).clone() for n in names]))
    side = engine.side_stream(head_eng.device)
    big = synth.waveforms(16, 64000, batch_idx=1).cuda()
    for rep in range(3):
        moved = []
        for i, f in enumerate(feats):
            torch.cuda.synchronize()
            trunk_eng.ssl(big)  # main stream: a long trunk
            with torch.cuda.stream(side):
                out = head_eng.head(f).clone()
                taps = [head_eng.tap(n).clone() for n in names]
            torch.cuda.synchronize()
            if not torch.equal(out, alone[i][0]):
                first = next((n for n, a, b in zip(names, taps, alone[i][1]) if not torch.equal(a, b)), "logits only")
                moved.append((i, float((out - alone[i][0]).abs().max()), first))
        print(f"head engine {hd}, a {td} trunk beside it, pass {rep}: batches whose logits moved (index, max |d|, first tap that moved): {moved}", flush=True)
    del head_eng, trunk_eng
