import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def sub_sd(z, prefix):
    """Arrays named '<prefix>sd.<key>' -> {key: tensor}."""
    p = prefix + "sd."
    return {k[len(p):]: torch.from_numpy(v) for k, v in z.items() if k.startswith(p)}


@pytest.fixture(scope="session")
def has_gpu():
    return torch.cuda.is_available()


def teacher_oracle(sd, wave):
    """The CPU oracle's side of teacher_conditioning (the expensive half: 24 fp32 transformer layers on the host), so that
    several engines / precisions can be gated against ONE oracle pass over the same utterances."""
    from oracle import models as om
    t_ref = {}
    ref = om.xlsr_aasist_forward(sd, wave, taps=t_ref)
    return ref, t_ref


def teacher_conditioning(sd, wave, eng, oracle=None):
    """XLSR_AASIST parity bookkeeping shared by the teacher tests.  Returns (ref, got, rows): oracle logits, the
    engine's logits, and per utterance
      backend   |engine - oracle back-end run on the ENGINE's own SSL features|: the back-end alone, same inputs;
      same_topk whether the REFERENCE MODEL keeps its decisions under the trunk's rounding: GraphPool keeps the top
                half of the nodes in descending score order and the branches are merged position by position
                (models/aasist_modules.py:330-336, models/xlsr_aasist.py:160-162), so the model is discontinuous where
                two kept scores nearly tie (4-s clips: 106 gaps per utterance, the smallest typically ~5e-6).  True =
                the oracle back-end picks the same node sequences on its own fp32 features and on the engine's;
      margin    the smallest deciding score gap (on either feature set), for the record.
    oracle: a teacher_oracle(sd, wave) result to reuse (None: computed here)."""
    from oracle import aasist as oa
    from oracle import models as om
    _ssl, head = om.split(sd)
    ref, t_ref = oracle if oracle is not None else teacher_oracle(sd, wave)
    eng.enable_taps()
    got = eng.forward(wave.cuda()).cpu()
    feats = eng.tap("ssl").cpu().reshape(t_ref["ssl"].shape)
    eng.enable_taps(False)
    t_mid = {}
    mid = oa.aasist_backend(head, feats, t_mid)
    rows = []
    for j in range(wave.shape[0]):
        same = all(torch.equal(t_ref["pool_idx"][p][j], t_mid["pool_idx"][p][j]) for p in t_ref["pool_idx"])
        margin = min(min(float(t_ref["pool_margin"][p][j]), float(t_mid["pool_margin"][p][j])) for p in t_ref["pool_margin"])
        rows.append(dict(same_topk=bool(same), margin=margin, dlogit=float((got[j] - ref[j]).abs().max()),
                         backend=float((got[j] - mid[j]).abs().max()),
                         feat_rel_l2=float((feats[j] - t_ref["ssl"][j]).norm() / t_ref["ssl"][j].norm())))
    return ref, got, rows
