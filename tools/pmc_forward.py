"""Two forwards of the bench workload (Conformer student, B = 64, 4 s clips) for
rocprofv3 --pmc passes (tools/pmc_traffic.sh): HBM traffic per kernel launch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import engine, synth  # noqa: E402

sd = synth.model_state_dict("ConformerModel", n_layers=6)
eng = engine.Engine("conformer", n_layers=6, dtype=os.environ.get("AFX_DTYPE", "fp16"))
eng.load_state_dict(sd)
wave = synth.waveforms(64, 64000).cuda()
for _ in range(3):
    eng.forward(wave)
torch.cuda.synchronize()
