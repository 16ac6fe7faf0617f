"""conf_chain_kernel time per launch for the library named by AFX_LIB (timing experiments with
-DCHAIN_DBG builds of afx_conformer_fused.hip; one process per library)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import engine, synth  # noqa: E402

sd = synth.model_state_dict("ConformerModel", n_layers=1)
eng = engine.Engine("conformer", n_layers=1, dtype="fp16")
eng.load_state_dict(sd)
wave = synth.waveforms(64, 64000).cuda()
for _ in range(3):
    eng.forward(wave)
eng.profile_begin()
for _ in range(10):
    eng.forward(wave)
c = eng.profile_end()[os.environ.get("AFX_CLASS", "conf_chain_kernel")]
print(f"{os.environ.get('AFX_LIB', 'default')[-18:]:20s} {c['ms'] / c['launches'] * 1e3:7.1f} us per launch", flush=True)
