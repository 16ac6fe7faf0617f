"""Diagnostic: scores against the CPU oracle on checkpoints with XLS-R-style outlier channels (afx.synth.with_outliers), per
precision and outlier gain -- how large the operand copies get, how far the scores move, and whether the overflow guard
(Engine.check_finite) fires where a copy leaves its format.  Small models (4-layer trunk, 1-s clips) so that the oracle
takes seconds."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import engine, synth  # noqa: E402
from afx._lib import AfxError  # noqa: E402
from oracle import models as om  # noqa: E402

torch.set_num_threads(16)
NL, B, L = 4, 4, 16000
wave = synth.waveforms(B, L, batch_idx=77)
for arch, oname, kw in (("conformer", "ConformerModel", dict(n_encoders=2)), ("xlsr_aasist", "XLSR_AASIST", dict(head_scale=1.5))):
    base = synth.model_state_dict(oname, n_layers=NL, **kw)
    for gain in (1.0, 10.0, 30.0, 100.0, 300.0, 1000.0, 3000.0):
        sd = synth.with_outliers(base, gain=gain) if gain != 1.0 else base
        taps = {}
        fwd = om.conformer_forward if arch == "conformer" else om.xlsr_aasist_forward
        ref = fwd(sd, wave, taps=taps)
        resid = max(float(v.abs().max()) for k, v in taps.items() if k.startswith("layer")) if any(k.startswith("layer") for k in taps) else float("nan")
        line = f"{arch:12s} gain {gain:6.0f}: oracle max |residual| {resid:9.3g}, |logit| {float(ref.abs().max()):7.3g}"
        for dtype in ("fp16", "fp16x3", "fp32"):
            eng = engine.Engine(arch, n_layers=NL, dtype=dtype, **({"conf_blocks": 2} if arch == "conformer" else {}))
            eng.load_state_dict(sd)
            got = eng.forward(wave.cuda()).cpu()
            try:
                eng.check_finite()
                loud = ""
            except AfxError as e:
                loud = " GUARD: " + str(e)[:60]
            d = (got - ref).abs().max().item()
            line += f" | {dtype}: {d:8.2e}{loud}"
            del eng
        print(line, flush=True)
