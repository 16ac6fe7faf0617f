"""Per-hop breakdown of the KV-cached streaming mode against its offline restatement (oracle/streaming.py): relative L2 error
of the feature window, |dlogit| end to end, and |dlogit| of the back-end alone (GPU back-end vs the oracle back-end on the
GPU's own window) -- the split tests/conftest.py::teacher_conditioning makes for the offline teacher."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import engine, synth  # noqa: E402
from afx.streaming import KVCachedScorer  # noqa: E402
from oracle import aasist, models, streaming as ostream  # noqa: E402

n_layers, S, hop, hops = 2, 3, 4000, 21
sd = synth.model_state_dict("XLSR_AASIST", n_layers=n_layers)
eng = engine.Engine("xlsr_aasist", n_layers=n_layers, dtype="fp16")
eng.load_state_dict(sd)
eng.enable_taps()
stream = torch.cat([synth.waveforms(S, hop, batch_idx=3000 + i) for i in range(hops)], dim=1)
wins = []
want, sizes = ostream.block_causal_scores(sd, stream, hop, windows=wins)
_ssl, head = models.split(sd)
sc = KVCachedScorer(eng, sd, S, window=64000, hop=hop)
for i in range(hops):
    got = sc.kv_logits = None
    s = sc.push(stream[:, i * hop:(i + 1) * hop].cuda())
    f = eng.tap("ssl").cpu().reshape(wins[i].shape)
    rel = ((f - wins[i]).norm() / wins[i].norm()).item()
    mid = aasist.aasist_backend(head, f)
    print(f"hop {i:2d} T {wins[i].shape[1]:3d}: feature rel L2 {rel:.2e}  |dscore| end-to-end {(s.cpu() - want[i][:, 1]).abs().max().item():.2e}  "
          f"back-end alone {(s.cpu() - mid[:, 1]).abs().max().item():.2e}")
