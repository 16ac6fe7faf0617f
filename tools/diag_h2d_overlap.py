"""Diagnostic: does the scoring loop's H2D prefetch (afx.harness.prefetch_to_device) really run under the forward?"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import engine, synth  # noqa: E402
from afx.harness import prefetch_to_device  # noqa: E402


def main():
    n = 30
    sd = synth.model_state_dict("ConformerModel", n_layers=6)
    eng = engine.Engine("conformer", n_layers=6, dtype="fp16")
    eng.load_state_dict(sd)
    wave = synth.waveforms(64, 64000, batch_idx=0).cuda()
    host = wave.cpu().pin_memory()

    def timed(name, fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n * 1e3
        print(f"{name:64s} {dt:7.3f} ms/step", flush=True)

    timed("resident, one stream", lambda: [eng.forward(wave) for _ in range(n)])
    timed("resident, two streams", lambda: ([eng.forward_overlapped(wave) for _ in range(n)], eng.join()))
    timed("H2D only, current stream (16.4 MB pinned)", lambda: [host.to("cuda", non_blocking=True) for _ in range(n)])
    side = torch.cuda.Stream()

    def h2d_side():
        for _ in range(n):
            with torch.cuda.stream(side):
                host.to("cuda", non_blocking=True)
        torch.cuda.current_stream().wait_stream(side)
    timed("H2D only, side stream", h2d_side)
    timed("H2D + forward, one stream", lambda: [eng.forward(host.to("cuda", non_blocking=True)) for _ in range(n)])
    timed("prefetch_to_device + forward", lambda: [eng.forward(x) for _m, x in prefetch_to_device(((i, host) for i in range(n)), "cuda")])
    timed("prefetch_to_device + forward_overlapped", lambda: ([eng.forward_overlapped(x) for _m, x in prefetch_to_device(((i, host) for i in range(n)), "cuda")], eng.join()))

    # the same prefetch with a static pair of device buffers and an explicit copy_ (no allocator, no record_stream)
    bufs = [torch.empty_like(wave), torch.empty_like(wave)]
    evs = [torch.cuda.Event(), torch.cuda.Event()]
    done = [torch.cuda.Event(), torch.cuda.Event()]

    def static_prefetch():
        cur = torch.cuda.current_stream()
        with torch.cuda.stream(side):
            bufs[0].copy_(host, non_blocking=True)
            evs[0].record(side)
        for i in range(n):
            k = i & 1
            if i + 1 < n:
                side.wait_event(done[k ^ 1]) if i >= 1 else None
                with torch.cuda.stream(side):
                    bufs[k ^ 1].copy_(host, non_blocking=True)
                    evs[k ^ 1].record(side)
            cur.wait_event(evs[k])
            eng.forward(bufs[k])
            done[k].record(cur)
    timed("two static buffers + copy_ on the side stream + forward", static_prefetch)


if __name__ == "__main__":
    main()
