"""Diagnostic: mhsa_kernel with V row-major in LDS + ds_read_b64_tr_b16 (knob mhsa_vtr = 1) against the V^T image: bits and time."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import kernels  # noqa: E402
from afx._lib import lib  # noqa: E402


def main():
    for B, T in ((64, 199), (16, 199), (64, 201), (3, 137)):
        qkv = (0.5 * torch.randn(B * T, 3072, device="cuda")).half()
        res = {}
        line = f"B={B:3d} T={T}:"
        for vtr in (0, 1, 0, 1):
            lib().afx_debug_set(b"mhsa_vtr", vtr)
            for _ in range(5):
                out = kernels.mhsa("fp16", qkv, B, T, 16)
            torch.cuda.synchronize()
            res.setdefault(vtr, out.clone())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                kernels.mhsa("fp16", qkv, B, T, 16)
            e1.record()
            torch.cuda.synchronize()
            line += f"  vtr={vtr} {e0.elapsed_time(e1) / 50 * 1e3:6.1f} us"
        line += f"  bit-identical: {torch.equal(res[0], res[1])}  max|d| {(res[0].float() - res[1].float()).abs().max().item():.2e}"
        print(line, flush=True)
    lib().afx_debug_set(b"mhsa_vtr", 0)


def end_to_end():
    """Both workloads in the scoring loop's two-stream form, knob off / on / off / on."""
    import time
    from afx import engine, synth
    for arch, oname, nl, B in (("conformer", "ConformerModel", 6, 64), ("xlsr_aasist", "XLSR_AASIST", 24, 16)):
        sd = synth.model_state_dict(oname, n_layers=nl)
        eng = engine.Engine(arch, n_layers=nl, dtype="fp16")
        eng.load_state_dict(sd)
        wave = synth.waveforms(B, 64000, batch_idx=0).cuda()
        line = f"{arch} B={B}:"
        for vtr in (0, 1, 0, 1):
            lib().afx_debug_set(b"mhsa_vtr", vtr)
            for _ in range(5):
                eng.forward_overlapped(wave)
            eng.join()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                eng.forward_overlapped(wave)
            eng.join()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 30
            line += f"  vtr={vtr} {dt * 1e3:.3f} ms ({B / dt:.0f} utt/s)"
        print(line, flush=True)
        del eng
    lib().afx_debug_set(b"mhsa_vtr", 1)


def zsplit():
    """Teacher B = 16: query tiles of a (utterance, head) on one workgroup (default at B x H >= 256) or split over two."""
    import time
    from afx import engine, synth
    sd = synth.model_state_dict("XLSR_AASIST", n_layers=24)
    eng = engine.Engine("xlsr_aasist", n_layers=24, dtype="fp16")
    eng.load_state_dict(sd)
    for B in (16, 8, 24):
        wave = synth.waveforms(B, 64000, batch_idx=0).cuda()
        line = f"xlsr_aasist B={B}:"
        ref = None
        for z in (0, 2, 0, 2):
            lib().afx_debug_set(b"mhsa_zsplit", z)
            for _ in range(5):
                out = eng.forward_overlapped(wave)
            eng.join()
            torch.cuda.synchronize()
            ref = out.clone() if ref is None else ref
            same = torch.equal(ref, out)
            t0 = time.perf_counter()
            for _ in range(30):
                eng.forward_overlapped(wave)
            eng.join()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 30
            line += f"  zsplit={z} {dt * 1e3:.3f} ms{'' if same else ' DIFF'}"
        print(line, flush=True)
    lib().afx_debug_set(b"mhsa_zsplit", 0)


def knob(name, values):
    """Teacher B = 16 and student B = 64 steps under an integer knob of afx_debug_set."""
    import time
    from afx import engine, synth
    for arch, oname, nl, B in (("xlsr_aasist", "XLSR_AASIST", 24, 16), ("conformer", "ConformerModel", 6, 64)):
        sd = synth.model_state_dict(oname, n_layers=nl)
        eng = engine.Engine(arch, n_layers=nl, dtype="fp16")
        eng.load_state_dict(sd)
        wave = synth.waveforms(B, 64000, batch_idx=0).cuda()
        line = f"{arch} B={B}:"
        ref = None
        for v in list(values) * 2:
            lib().afx_debug_set(name.encode(), v)
            for _ in range(5):
                out = eng.forward_overlapped(wave)
            eng.join()
            torch.cuda.synchronize()
            ref = out.clone() if ref is None else ref
            same = torch.equal(ref, out)
            t0 = time.perf_counter()
            for _ in range(30):
                eng.forward_overlapped(wave)
            eng.join()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 30
            line += f"  {name}={v} {dt * 1e3:.3f} ms{'' if same else ' DIFF'}"
        print(line, flush=True)
        del eng


if __name__ == "__main__":
    if "knob" in sys.argv:
        i = sys.argv.index("knob")
        knob(sys.argv[i + 1], [int(x) for x in sys.argv[i + 2:]])
        sys.exit(0)
    zsplit() if "zsplit" in sys.argv else (end_to_end() if "e2e" in sys.argv else main())
