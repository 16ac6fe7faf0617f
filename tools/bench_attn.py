"""A/B of the attention kernels' waves per workgroup (mhsa_waves / conf_attn_waves 4 / 7) at the student's (B = 64) and the teacher's
(B = 16, query tiles split over two workgroups) shapes, interleaved rounds."""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import kernels as K  # noqa: E402
from afx._lib import check, lib  # noqa: E402


def timeit(fn, reps=20):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    g = torch.Generator(device="cuda").manual_seed(1)
    for B, T in ((64, 199), (16, 199), (64, 201), (256, 149)):
        qkv = torch.randn(B * T, 3 * 16 * 64, generator=g, device="cuda").half()
        times = {4: [], 7: []}
        outs = {}
        for _ in range(5):
            for nw in (4, 7):
                check(lib().afx_debug_set(b"mhsa_waves", nw))
                times[nw].append(timeit(lambda: K.mhsa("fp16", qkv, B, T, 16)))
                outs[nw] = K.mhsa("fp16", qkv, B, T, 16)
        check(lib().afx_debug_set(b"mhsa_waves", 7))
        same = torch.equal(outs[4], outs[7])
        print(f"mhsa B={B} T={T}: 4 waves {statistics.median(times[4]):6.1f} us  7 waves {statistics.median(times[7]):6.1f} us  identical={same}", flush=True)
    for B, N in ((64, 200), (16, 200), (256, 150)):
        q = torch.randn(B * N, 144, generator=g, device="cuda")
        kv = torch.randn(B * N, 288, generator=g, device="cuda")
        rel = torch.randn(1025, 36, generator=g, device="cuda") * 0.1
        times = {4: [], 7: []}
        outs = {}
        for _ in range(5):
            for nw in (4, 7):
                check(lib().afx_debug_set(b"conf_attn_waves", nw))
                times[nw].append(timeit(lambda: K.conf_attn_mfma("fp16", q, kv, rel, B, N, 4, 36)))
                outs[nw] = K.conf_attn_mfma("fp16", q, kv, rel, B, N, 4, 36)
        check(lib().afx_debug_set(b"conf_attn_waves", 7))
        print(f"conf_attn_mfma B={B} N={N} (incl. table pack): 4 waves {statistics.median(times[4]):6.1f} us  7 waves {statistics.median(times[7]):6.1f} us  "
              f"identical={torch.equal(outs[4], outs[7])}", flush=True)


if __name__ == "__main__":
    main()
