"""Oracle: steps either side of the model (TEST INFRASTRUCTURE).

* pre_emphasis   -- data/preprocess.py:16-29  (reflect pad 1 left, conv [-c, 1])
* adjust_duration -- data/test_set.py:201-227 (tile then crop the first N)
* pad_tile       -- data/test_set.py:139-146 (np.tile variant, 64 600 samples)
* eer_percent    -- trainer.py:134-139       (brentq on the interpolated ROC)

Their host modules import torchaudio / wandb (absent), so the bodies are
restated; known-answer vectors live in tests/golden/pre_eer.npz.
"""
import numpy as np
import torch


def pre_emphasis(x, coef=0.97):
    """x (B,L) -> (B,L):  y[0] = x[0] - c*x[1];  y[t] = x[t] - c*x[t-1].
    (Unlike the reference's ``.squeeze()`` the batch axis is kept at B=1.)"""
    prev = torch.cat([x[:, 1:2], x[:, :-1]], dim=1)
    return x - coef * prev


def adjust_duration(x, duration):
    """1-D waveform -> exactly ``duration`` samples from the start, short clips
    repeated whole plus a residue (data/test_set.py:201-227)."""
    x = x.reshape(-1)
    n = x.shape[0]
    if n < duration:
        parts = [x] * (duration // n)
        if duration % n > 0:
            parts.append(x[: duration % n])
        x = torch.cat(parts, dim=0)
    return x[:duration]


def pad_tile(x, max_len=64600):
    """data/test_set.py:139-146."""
    n = x.shape[0]
    if n >= max_len:
        return x[:max_len]
    reps = int(max_len / n) + 1
    return np.tile(x, (1, reps))[:, :max_len][0]


def eer_percent(scores, labels):
    """trainer.py:134-139."""
    from scipy.interpolate import interp1d
    from scipy.optimize import brentq
    from sklearn import metrics
    fpr, tpr, _ = metrics.roc_curve(labels, scores, pos_label=1)
    return brentq(lambda x: 1.0 - x - interp1d(fpr, tpr)(x), 0.0, 1.0) * 100
