"""Attribution of mhsa_kernel's time (timing-only builds of afx_attn.hip with -DMHSA_DBG=bits, WRONG results; loaded through
AFX_LIB): one (B, 199 frames, 16 heads) launch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import kernels  # noqa: E402


def main():
    tag = sys.argv[1]
    line = f"{tag:44s}"
    for B in (64, 16):
        qkv = (0.5 * torch.randn(B * 199, 3072, device="cuda")).half()
        for _ in range(5):
            kernels.mhsa("fp16", qkv, B, 199, 16)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            kernels.mhsa("fp16", qkv, B, 199, 16)
        e1.record()
        torch.cuda.synchronize()
        line += f"  B={B}: {e0.elapsed_time(e1) / 50 * 1e3:6.1f} us"
    print(line, flush=True)


if __name__ == "__main__":
    main()
