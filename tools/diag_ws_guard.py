"""Diagnostic: does any kernel of a forward write OUTSIDE the workspace it was given?  The workspace is a window inside a larger
buffer whose margins hold a byte pattern; after the forward the margins must still hold it."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import engine, synth  # noqa: E402
from afx._lib import lib  # noqa: E402

G = 64 << 20
for arch, oname, kw, ekw in (("xlsr_aasist", "XLSR_AASIST", dict(head_scale=1.5), {}), ("conformer", "ConformerModel", dict(n_encoders=2), dict(conf_blocks=2))):
    for dtype in ("fp16x3", "fp16", "fp32"):
        sd = synth.model_state_dict(oname, n_layers=2, **kw)
        eng = engine.Engine(arch, n_layers=2, dtype=dtype, **ekw)
        eng.load_state_dict(sd)
        for B, L in ((5, 16000), (3, 16000), (7, 16000), (16, 64000)):
            wave = synth.waveforms(B, L, batch_idx=3).cuda()
            n = lib().afx_workspace_bytes(eng._h, B, L)
            big = torch.full((G + n + G,), 0x5A, dtype=torch.uint8, device="cuda")
            eng._ws = big[G:G + n]
            eng.forward(wave)
            torch.cuda.synchronize()
            lo = (big[:G] != 0x5A).nonzero()
            hi = (big[G + n:] != 0x5A).nonzero()
            msg = "clean"
            if lo.numel() or hi.numel():
                msg = (f"WRITES OUTSIDE: {lo.numel()} bytes before the workspace (nearest at -{G - int(lo.max()) if lo.numel() else 0}), "
                       f"{hi.numel()} bytes behind it (first at +{int(hi.min()) if hi.numel() else 0}, last at +{int(hi.max()) if hi.numel() else 0})")
            print(f"{arch} {dtype} B {B} L {L}: workspace {n} bytes: {msg}", flush=True)
            eng._ws = None
            del big
        del eng
        torch.cuda.empty_cache()
