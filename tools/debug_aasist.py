"""GPU-side stage-by-stage comparison of the AASIST back-end with the CPU oracle
(diagnostic; prints max abs differences per intermediate)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import engine, synth  # noqa: E402
from oracle import aasist as oa  # noqa: E402

z = dict(np.load(os.path.join(ROOT, "tests/golden/aasist_backend.npz")))
head = {k[3:]: torch.from_numpy(v) for k, v in z.items() if k.startswith("sd.")}
sd = dict(synth.ssl_state_dict(1))
sd.update(head)
eng = engine.Engine("xlsr_aasist", n_layers=1, dtype="fp16")
eng.load_state_dict(sd)
eng.enable_taps()
for tag in ("t199", "t49"):
    feats = torch.from_numpy(z[tag + ".feats"])
    B = feats.shape[0]
    got = eng.head(feats.cuda()).cpu()
    e_S, e_T = oa.aasist_front(head, feats)
    gS = oa.graph_attention(head, "GAT_layer_S.", e_S, 2.0)
    gT = oa.graph_attention(head, "GAT_layer_T.", e_T, 2.0)
    oS = oa.graph_pool(head, "pool_S.", gS, 0.5)
    oT = oa.graph_pool(head, "pool_T.", gT, 0.5)
    T1, S1, m1 = oa.htrg_graph_attention(head, "HtrgGAT_layer_ST11.", oT, oS, head["master1"], 100.0)
    S1p = oa.graph_pool(head, "pool_hS1.", S1, 0.5)
    T1p = oa.graph_pool(head, "pool_hT1.", T1, 0.5)
    Ta, Sa, ma = oa.htrg_graph_attention(head, "HtrgGAT_layer_ST12.", T1p, S1p, m1, 100.0)
    ref = {"e_S": e_S, "e_T": e_T, "gat_S": gS, "gat_T": gT, "out_S": oS, "out_T": oT, "b1_T1": T1, "b1_S1": S1,
           "b1_m1": m1, "b1_T1p": T1p, "b1_S1p": S1p, "b1_Ta": Ta, "b1_Sa": Sa, "b1_ma": ma}
    print("==", tag, "logits", got.tolist(), "golden", z[tag + ".logits"].tolist())
    for k, v in ref.items():
        g = eng.tap(k).cpu().reshape(v.shape)
        d = (g - v).abs()
        rows = d.reshape(-1, v.shape[-1]).max(dim=1)[0]
        bad = (rows > 1e-3).nonzero().reshape(-1).tolist()
        print(f"{k:8s} shape {tuple(v.shape)} max|d| {d.max().item():.3e}  rows>1e-3: {bad[:12]}")

# ---- whole teacher model (2-layer trunk): where does the logit error come from? ----
from oracle import models as om  # noqa: E402
from oracle import ssl_trunk  # noqa: E402
sd = synth.model_state_dict("XLSR_AASIST", n_layers=2)
wave = synth.waveforms(5, 64000, batch_idx=2)
taps = {}
ref = om.xlsr_aasist_forward(sd, wave, taps=taps)
eng2 = engine.Engine("xlsr_aasist", n_layers=2, dtype="fp16")
eng2.load_state_dict(sd)
eng2.enable_taps()
got = eng2.forward(wave.cuda()).cpu()
print("== teacher e2e: max|dlogit|", (got - ref).abs().max().item())
print("got", got.tolist())
print("ref", ref.tolist())
for k in ("ssl", "e_S", "e_T", "out_S", "out_T", "hidden"):
    g = eng2.tap(k).cpu().reshape(taps[k].shape)
    d = (g - taps[k]).abs()
    print(f"{k:8s} max|d| {d.max().item():.3e} rel {(g - taps[k]).norm().item() / taps[k].norm().item():.3e} absmax {taps[k].abs().max().item():.3f}")
# head alone on the ORACLE's ssl features: isolates the back-end from trunk rounding
got_h = eng2.head(taps["ssl"].cuda()).cpu()
print("head on oracle feats: max|dlogit|", (got_h - ref).abs().max().item())
for k in ("e_S", "e_T", "out_S", "out_T", "hidden"):
    g = eng2.tap(k).cpu().reshape(taps[k].shape)
    print(f"  {k:8s} max|d| {(g - taps[k]).abs().max().item():.3e}")
