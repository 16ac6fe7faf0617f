"""Score parity of the default (fp16) engine against the CPU oracle on a larger sample than bench.py's:
N utterances of 4 s, Conformer student (BASELINE config 2 shape).  Prints the distribution of |dlogit|."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import engine, synth  # noqa: E402
from oracle import models  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
sd = synth.model_state_dict("ConformerModel", n_layers=6)
eng = engine.Engine("conformer", n_layers=6, dtype=os.environ.get("AFX_DTYPE", "fp16"))
eng.load_state_dict(sd)
errs = []
for i in range(0, n, 8):
    wave = synth.waveforms(8, 64000, batch_idx=1000 + i)
    ref = models.conformer_forward(sd, wave)
    got = eng.forward(wave.cuda()).cpu()
    errs.append((got - ref).abs().max(dim=1)[0])
    print(f"utterances {i:3d}-{i + 7:3d}: max |dlogit| {errs[-1].max().item():.2e}", flush=True)
e = torch.cat(errs)
print(f"{n} utterances: max {e.max().item():.2e}  mean {e.mean().item():.2e}  median {e.median().item():.2e}  "
      f"over 1e-3: {(e > 1e-3).sum().item()}")
