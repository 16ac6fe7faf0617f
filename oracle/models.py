"""Oracle: whole-model forwards keyed by the reference class names
(TEST INFRASTRUCTURE).  State dicts use the reference checkpoint key names:
``ssl_model.model.<fairseq names>`` + the head's own keys (optionally with the
``module.`` prefix that utils.py:13-43 adds/strips).
"""
import torch

from . import aasist, conformer, ssl_trunk

SSL_PREFIX = "ssl_model.model."


def strip_module(sd):
    """utils.py:13-43 (data_parallel=False branch)."""
    return {(k[7:] if k.startswith("module") else k): v for k, v in sd.items()}


def split(sd):
    sd = strip_module(sd)
    ssl = {k[len(SSL_PREFIX):]: v for k, v in sd.items() if k.startswith(SSL_PREFIX)}
    head = {k: v for k, v in sd.items() if not k.startswith(SSL_PREFIX)}
    return ssl, head


@torch.no_grad()
def xlsr_aasist_forward(sd, wave, q=None, taps=None, heads=16):
    """models/xlsr_aasist.py:86-177 (XLSR_AASIST / My_XLSR_AASIST)."""
    ssl, head = split(sd)
    feats = ssl_trunk.ssl_forward(ssl, wave.squeeze(-1) if wave.ndim == 3 else wave, heads=heads, q=q, taps=taps)
    if taps is not None:
        taps["ssl"] = feats
    return aasist.aasist_backend(head, feats, taps)


@torch.no_grad()
def conformer_forward(sd, wave, heads=4, q=None, taps=None, ssl_heads=16):
    """models/conformer_baseline.py:54-64 (Model) / :88-99 (MyModel, Q4 fixed)."""
    ssl, head = split(sd)
    feats = ssl_trunk.ssl_forward(ssl, wave.squeeze(-1) if wave.ndim == 3 else wave, heads=ssl_heads, q=q, taps=taps)
    if taps is not None:
        taps["ssl"] = feats
    return conformer.conformer_head(head, feats, heads=heads, q=q, taps=taps)


@torch.no_grad()
def ssl_forward(sd, wave, q=None, heads=16):
    """models/fe.py:17-24 / models/models.py:23-29 (feature extractor alone)."""
    sd = strip_module(sd)
    if any(k.startswith("model.") for k in sd):
        sd = {k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")}
    return ssl_trunk.ssl_forward(sd, wave, heads=heads, q=q)
