"""Where the fp16 engine's score error comes from: trunk vs Conformer head (exact mode as the reference)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import engine, synth  # noqa: E402

sd = synth.model_state_dict("ConformerModel", n_layers=6)
ex = engine.Engine("conformer", n_layers=6, dtype="fp32")
ex.load_state_dict(sd)
hf = engine.Engine("conformer", n_layers=6, dtype="fp16")
hf.load_state_dict(sd)
ex.enable_taps()
hf.enable_taps()
wave = synth.waveforms(16, 64000, batch_idx=1000).cuda()
ref = ex.forward(wave)
f_ex = ex.tap("ssl").reshape(16, 199, 1024)
got = hf.forward(wave)
f_hf = hf.tap("ssl").reshape(16, 199, 1024)
d = lambda a, b: (a - b).abs().max(dim=1)[0]
print("fp16 end to end vs exact:          max %.2e mean %.2e" % (d(got, ref).max(), d(got, ref).mean()))
print("signed mean of (fp16 - exact):     ", (got - ref).mean(dim=0).tolist())
h1 = hf.head(f_ex)
print("exact trunk features -> fp16 head: max %.2e mean %.2e" % (d(h1, ref).max(), d(h1, ref).mean()))
h2 = ex.head(f_hf)
print("fp16 trunk features -> exact head: max %.2e mean %.2e" % (d(h2, ref).max(), d(h2, ref).mean()))
print("trunk feature rel L2 error: %.2e" % ((f_hf - f_ex).norm() / f_ex.norm()).item())
for name in ("tokens", "block0", "block1", "block2", "block3"):
    a, b = hf.tap(name), ex.tap(name)
    print(f"  {name}: rel L2 {((a - b).norm() / b.norm()).item():.2e}")
