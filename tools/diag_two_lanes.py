"""Another way of keeping two batches in flight: whole forwards of consecutive batches on ALTERNATING streams (each with its own workspace),
instead of the back-end of batch i beside the trunk of batch i+1.  The teacher's small-M products fill 200 of 256 CUs; a second batch's
kernels can take the rest.  Times per batch over 24 batches, logits against the one-stream forward, bit for bit.
    python tools/diag_two_lanes.py [dtype]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import engine, synth  # noqa: E402
from afx._lib import check, lib, ptr  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "fp16"
l = lib()
for arch, oname, nl, B in (("xlsr_aasist", "XLSR_AASIST", 24, 16), ("conformer", "ConformerModel", 6, 64)):
    eng = engine.Engine(arch, n_layers=nl, dtype=dtype)
    eng.load_state_dict(synth.model_state_dict(oname, n_layers=nl, **({"head_scale": 1.5} if arch == "xlsr_aasist" else {})))
    L, NB = 64000, 24
    waves = [synth.waveforms(B, L, batch_idx=40 + i).cuda() for i in range(4)]
    want = [eng.forward(w).clone() for w in waves]
    nbytes = l.afx_workspace_bytes(eng._h, B, L)

    def timed(fn, join):
        for _ in range(4):
            fn(0)
        join()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        outs = [fn(i) for i in range(NB)]
        join()
        e1.record()
        torch.cuda.synchronize()
        same = all(torch.equal(o, want[i % 4]) for i, o in enumerate(outs))
        return e0.elapsed_time(e1) / NB, same

    ms, same = timed(lambda i: eng.forward(waves[i % 4]), lambda: None)
    print(f"{arch} {dtype} B {B}: one stream {ms:.3f} ms per batch ({B * 1e3 / ms:.0f} utt/s), same bits {same}", flush=True)
    ms, same = timed(lambda i: eng.forward_overlapped(waves[i % 4]), eng.join)
    print(f"{arch} {dtype} B {B}: back-end beside the next trunk {ms:.3f} ms per batch ({B * 1e3 / ms:.0f} utt/s), same bits {same}", flush=True)
    eng.set_issue("lanes")
    ms, same = timed(lambda i: eng.forward_overlapped(waves[i % 4]), eng.join)
    print(f"{arch} {dtype} B {B}: Engine.forward_lanes (current stream + side stream) {ms:.3f} ms per batch ({B * 1e3 / ms:.0f} utt/s), same bits {same}", flush=True)
    eng.set_issue("overlap")
    for lanes in (2, 3):
        cur = torch.cuda.current_stream()
        streams = [torch.cuda.Stream() for _ in range(lanes)]
        wss = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(lanes)]
        outs_buf = [[torch.empty(B, 2, device="cuda") for _ in range(NB + 8)] for _ in range(1)]
        cnt = [0]

        def fn(i):
            k = cnt[0] % lanes
            out = outs_buf[0][cnt[0] % (NB + 8)]
            cnt[0] += 1
            streams[k].wait_stream(cur)  # (the inputs are ready on the current stream)
            check(l.afx_forward(eng._h, ptr(waves[i % 4]), B, L, ptr(out), ptr(wss[k]), nbytes, C.c_void_p(streams[k].cuda_stream)))
            return out

        def join():
            for st in streams:
                cur.wait_stream(st)
        ms, same = timed(fn, join)
        print(f"{arch} {dtype} B {B}: {lanes} whole forwards in flight on {lanes} streams {ms:.3f} ms per batch ({B * 1e3 / ms:.0f} utt/s), same bits {same}", flush=True)
    del eng
    torch.cuda.empty_cache()
