"""Diagnostic: does a forward depend on what its workspace held before (it must not), and does the two-stream form (back-end on
the side stream under the next trunk) give the one-stream form's bits -- also when every call is followed by a device
synchronise (then nothing runs concurrently: a difference that vanishes is a race between the two streams' kernels)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import engine, synth  # noqa: E402

for arch, oname, kw, ekw in (("xlsr_aasist", "XLSR_AASIST", dict(head_scale=1.5), {}), ("conformer", "ConformerModel", dict(n_encoders=2), dict(conf_blocks=2))):
    for dtype in ("fp16x3", "fp32", "fp16"):
        sd = synth.model_state_dict(oname, n_layers=2, **kw)
        eng = engine.Engine(arch, n_layers=2, dtype=dtype, **ekw)
        eng.load_state_dict(sd)
        waves = [synth.waveforms(b, 16000, batch_idx=700 + i).cuda() for i, b in enumerate([5, 5, 5, 3, 7, 5, 1])]
        want = [eng.forward(w).clone() for w in waves]
        feats = [eng.ssl(w).clone() for w in waves] if arch == "xlsr_aasist" else None
        for fill in (0xFF, 0x00):
            bad = []
            for i, w in enumerate(waves):
                eng._ws.fill_(fill)
                got = eng.forward(w)
                if not torch.equal(got, want[i]):
                    bad.append((i, float((got - want[i]).abs().max())))
            print(f"{arch} {dtype}: workspace pre-filled with 0x{fill:02X}: batches that changed {bad}", flush=True)
        for sync in (False, True):
            got = []
            for w in waves:
                got.append(eng.forward_overlapped(w))
                if sync:
                    torch.cuda.synchronize()
            eng.join()
            torch.cuda.synchronize()
            diff = [(i, float((g - w_).abs().max())) for i, (g, w_) in enumerate(zip(got, want)) if not torch.equal(g, w_)]
            print(f"{arch} {dtype}: two-stream{' + synchronise after every call' if sync else ''} vs one-stream, batches that differ: {diff}", flush=True)
        if feats is not None:  # the trunk alone, with the OTHER model's head running beside it on the side stream
            side = engine.side_stream(eng.device)
            eng2 = engine.Engine(arch, n_layers=2, dtype=dtype, **ekw)  # (its own workspace: nothing shared with `eng` but the GPU)
            eng2.load_state_dict(sd)
            eng2.head(feats[0])
            torch.cuda.synchronize()
            bad = []
            for i, w in enumerate(waves):
                with torch.cuda.stream(side):
                    eng2.head(feats[(i + 1) % len(feats)])
                f = eng.ssl(w)
                torch.cuda.synchronize()
                if not torch.equal(f, feats[i]):
                    bad.append((i, float((f - feats[i]).abs().max())))
            print(f"{arch} {dtype}: SSL features with a back-end running beside the trunk, batches that changed: {bad}", flush=True)
        del eng
