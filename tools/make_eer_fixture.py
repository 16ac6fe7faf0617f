"""Oracle scores of the BENCHMARKED student (BASELINE configs[1]: first-6 XLS-R trunk + 4 Conformer blocks, emb 144) on
N = 64 x EER_BATCHES (default 64 -> 4096, SURVEY 8d's trial count) four-second synthetic trials, for the EER-delta /
parity-sweep test (tests/test_gpu_models.py).  CPU only (the oracle is ~5-10 utterances/s: 12 minutes on 8 cores); writes
tests/golden/eer_student_<N>.npz = {scores (N,) fp32 bonafide logits, logits (N,2), labels (N,) drawn from the oracle's
scores plus noise so that its EER sits at 10-20 %}.  The waveforms are synth.waveforms(64, 64000, batch_idx=9000 + i),
i < EER_BATCHES: the test regenerates them."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import synth  # noqa: E402
from oracle import models, pre  # noqa: E402

N_BATCH, B = int(os.environ.get("EER_BATCHES", 64)), 64
sd = synth.model_state_dict("ConformerModel", n_layers=6)
out = []
t0 = time.time()
for i in range(N_BATCH):
    wave = synth.waveforms(B, 64000, batch_idx=9000 + i)
    out.append(torch.cat([models.conformer_forward(sd, wave[j:j + 16]) for j in range(0, B, 16)]))
    print(f"batch {i + 1}/{N_BATCH}  {time.time() - t0:.0f} s", flush=True)
logits = torch.cat(out)
ref = logits[:, 1]
g = torch.Generator().manual_seed(4096)
noisy = ref + ref.std() * 0.8 * torch.randn(ref.shape, generator=g)
labels = (noisy > noisy.median()).long().numpy()
eer = pre.eer_percent(ref.numpy(), labels)
print(f"oracle EER {eer:.4f} %")
np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"eer_student_{N_BATCH * B}.npz"), scores=ref.numpy().astype(np.float32),
                    logits=logits.numpy().astype(np.float32), labels=labels.astype(np.int8), eer=np.float64(eer))
