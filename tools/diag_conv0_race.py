"""Diagnostic: conv layer 0 in its fp32 form (conv0_kernel<F32T>: what dtype "fp32" and "fp16x3" engines run), launched over
and over on the main stream while an AASIST back-end runs on the side stream -- of the SAME engine object or of another one --
or while nothing else runs.  Every output must equal the first one bit for bit."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import engine, kernels as K, synth  # noqa: E402

sd = synth.model_state_dict("XLSR_AASIST", n_layers=2, head_scale=1.5)
P = "ssl_model.model.feature_extractor.conv_layers.0."
w, b, g, be = (sd[P + k].cuda() for k in ("0.weight", "0.bias", "2.1.weight", "2.1.bias"))
wave = synth.waveforms(5, 16000, batch_idx=701).cuda()
ref = K.conv0("fp32", wave, w.reshape(512, 10), b, g, be).clone()
side = engine.side_stream(torch.device("cuda", 0))
for hd in ("fp16x3", "fp16"):
    eng = engine.Engine("xlsr_aasist", n_layers=2, dtype=hd)
    eng.load_state_dict(sd)
    feats = eng.ssl(wave).clone()
    for mode in ("nothing beside", "AASIST back-end beside"):
        bad = 0
        for it in range(40):
            torch.cuda.synchronize()
            if mode != "nothing beside":
                with torch.cuda.stream(side):
                    eng.head(feats)
            for _ in range(4):
                out = K.conv0("fp32", wave, w.reshape(512, 10), b, g, be)
                if not torch.equal(out, ref):
                    bad += 1
                    rows = (out != ref).reshape(-1, 512).any(dim=1).nonzero().flatten().tolist()
                    if bad <= 3:
                        print(f"   iteration {it}: rows that differ {rows[:6]} max |d| {float((out - ref).abs().max()):.3e}", flush=True)
        print(f"back-end engine {hd}, {mode}: {bad} of 160 conv0 launches differ from the first", flush=True)
    del eng
