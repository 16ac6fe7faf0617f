"""Kernel-class A/B (AFX_CLASS, default conf_chain_kernel) for a library variant (AFX_LIB): per-launch time from the engine's hipEvent classes and
the student's logits on a fixed batch (printed so that two processes can be compared)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import engine, synth  # noqa: E402

sd = synth.model_state_dict("ConformerModel", n_layers=1)
eng = engine.Engine("conformer", n_layers=1, dtype="fp16")
eng.load_state_dict(sd)
for B, L in ((64, 64000), (64, 16000), (7, 64000)):
    wave = synth.waveforms(B, L).cuda()
    for _ in range(3):
        out = eng.forward(wave)
    eng.profile_begin()
    for _ in range(10):
        eng.forward(wave)
    c = eng.profile_end()[os.environ.get("AFX_CLASS", "conf_chain_kernel")]
    print(f"{os.environ.get('AFX_LIB', 'default')[-16:]:18s} B={B:3d} L={L:6d}: {c['ms'] / c['launches'] * 1e3:7.1f} us per launch; "
          f"logit checksum {out.double().sum().item():+.9f} first {out[0].tolist()}", flush=True)
