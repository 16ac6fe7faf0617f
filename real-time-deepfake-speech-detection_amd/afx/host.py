"""Host-side plumbing shared by the drop-in ``models`` package: parameter containers
with the reference's state_dict key names, and the mixin that routes ``forward`` to
the native engine.  No arithmetic of the path happens here."""
import os

import torch
from torch import nn

from . import synth
from .engine import DEFAULT_DTYPE, Engine

CONV_LAYERS = [(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)] * 2


# ---- fairseq-named containers ---------------------------------
class _WeightNormConv(nn.Module):
    """Holds weight_g / weight_v / bias of the weight-normed positional conv."""

    def __init__(self, dim, groups, k):
        super().__init__()
        self.weight_g = nn.Parameter(torch.ones(1, 1, k))
        self.weight_v = nn.Parameter(torch.zeros(dim, dim // groups, k))
        self.bias = nn.Parameter(torch.zeros(dim))


class _SelfAttn(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.k_proj, self.v_proj, self.q_proj, self.out_proj = (nn.Linear(d, d) for _ in range(4))


class EncoderLayer(nn.Module):
    """fairseq TransformerSentenceEncoderLayer parameter set."""

    def __init__(self, d=1024, f=4096):
        super().__init__()
        self.self_attn = _SelfAttn(d)
        self.self_attn_layer_norm = nn.LayerNorm(d)
        self.fc1 = nn.Linear(d, f)
        self.fc2 = nn.Linear(f, d)
        self.final_layer_norm = nn.LayerNorm(d)


class _Encoder(nn.Module):
    def __init__(self, n_layers, d, f):
        super().__init__()
        self.pos_conv = nn.Sequential(_WeightNormConv(d, 16, 128))
        self.layers = nn.ModuleList([EncoderLayer(d, f) for _ in range(n_layers)])
        self.layer_norm = nn.LayerNorm(d)


class _FeatureExtractor(nn.Module):
    def __init__(self):
        super().__init__()
        blocks, cin = [], 1
        for c, k, s in CONV_LAYERS:  # keys conv_layers.{i}.0.* and conv_layers.{i}.2.1.*
            blocks.append(nn.Sequential(nn.Conv1d(cin, c, k, stride=s, bias=True), nn.Identity(),
                                        nn.Sequential(nn.Identity(), nn.LayerNorm(c), nn.Identity()), nn.Identity()))
            cin = c
        self.conv_layers = nn.ModuleList(blocks)


class Wav2Vec2Trunk(nn.Module):
    """Parameter container with fairseq Wav2Vec2Model's names for the sub-graph that
    ``forward(x, mask=False, features_only=True)['x']`` uses.  Weights start from the
    seeded synthetic generator (no checkpoint can be fetched offline)."""

    def __init__(self, n_layers=24, d=1024, f=4096):
        super().__init__()
        with torch.device("meta"):  # shapes only; the values come from the seeded generator below
            self.feature_extractor = _FeatureExtractor()
            self.layer_norm = nn.LayerNorm(512)
            self.post_extract_proj = nn.Linear(512, d)
            self.encoder = _Encoder(n_layers, d, f)
        self.load_state_dict(synth.ssl_state_dict(n_layers, prefix=""), strict=True, assign=True)

    def forward(self, source, mask=False, features_only=True, **kw):
        raise RuntimeError("Wav2Vec2Trunk holds parameters only; call the owning XLSR_FE / model")


class _StubPickle:
    """``pickle_module`` for torch.load of a fairseq checkpoint (``xlsr2_300m.pt`` carries argparse / omegaconf /
    fairseq config objects beside the tensors).  ALLOW-LIST of exact ``(module, name)`` pairs: only what rebuilding
    tensors and plain containers needs is resolved to the real object -- torch's tensor / parameter rebuild helpers, the
    typed storage classes, ``torch.Size`` / ``torch.device`` / the dtype singletons, ``collections.OrderedDict``,
    ``argparse.Namespace``, numpy's array / scalar reconstructors and a few builtin containers.  The pair is looked up
    with ONE ``getattr`` on the already-imported module, never through ``pickle``'s own ``find_class``: protocol >= 4
    resolves dotted names attribute by attribute, so "every attribute of torch.serialization" (round 2's rule) reached
    ``torch.serialization.os.system``, ``torch.storage.io.open`` and ``torch._utils._import_dotted_name``.  A name with
    a dot in it is never resolved.  EVERY other global -- installed or not -- becomes an inert stand-in class whose
    construction and ``__setstate__`` only store what they are given.  So a crafted file cannot run code through this
    loader; only the ``model`` state_dict is read afterwards."""
    import pickle as _pickle

    __name__ = "afx_stub_pickle"
    _STORAGES = ("UntypedStorage", "FloatStorage", "HalfStorage", "BFloat16Storage", "DoubleStorage", "LongStorage",
                 "IntStorage", "ShortStorage", "CharStorage", "ByteStorage", "BoolStorage", "ComplexFloatStorage",
                 "ComplexDoubleStorage")
    _ALLOWED = {("collections", "OrderedDict"), ("argparse", "Namespace"), ("torch", "Size"), ("torch", "device"),
                ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"),
                ("torch._utils", "_rebuild_parameter"), ("torch.nn.parameter", "Parameter"), ("torch", "Tensor"),
                ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
                ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"), ("numpy", "ndarray"),
                ("numpy", "dtype"), ("builtins", "set"), ("builtins", "frozenset"), ("builtins", "slice"),
                ("builtins", "complex"), ("builtins", "bytearray")} | {("torch", s) for s in _STORAGES}

    @staticmethod
    def _resolve(module, name):
        """The real object behind an allow-listed pair, or None."""
        import importlib
        import sys
        if "." in name:
            return None
        ok = (module, name) in _StubPickle._ALLOWED or \
            (module == "torch" and isinstance(torch.__dict__.get(name), torch.dtype))
        if not ok:
            return None
        try:
            mod = sys.modules.get(module) or importlib.import_module(module)
        except ImportError:
            return None
        return mod.__dict__.get(name)  # the module's OWN attribute: no attribute walk, no __getattr__ hook

    class Unpickler(_pickle.Unpickler):
        def find_class(self, module, name):
            obj = _StubPickle._resolve(module, name)
            if obj is not None:
                return obj

            def __init__(self, *a, **k):
                pass

            def __setstate__(self, state):
                self.__dict__.update(state if isinstance(state, dict) else {"state": state})
            return type(name.rpartition(".")[2] or "stub", (), {"__module__": module, "__init__": __init__,
                                                                "__setstate__": __setstate__,
                                                                "__call__": lambda self, *a, **k: None})

    @staticmethod
    def load(f, **kw):
        return _StubPickle.Unpickler(f, **kw).load()


_UNUSED_SSL = ("quantizer.", "project_q.", "final_proj.", "mask_emb", "target_glu.", "bn1.")


def load_ssl_checkpoint(trunk, path):
    """Load a fairseq ``xlsr2_300m.pt`` (``{"cfg"/"args": ..., "model": state_dict}``) or a plain
    state_dict file into the trunk container -- without fairseq (models/fe.py:11-14 uses
    ``fairseq.checkpoint_utils.load_model_ensemble_and_task``): config objects of absent packages are
    stubbed while unpickling, only the tensors are used.  Keys may carry the ``module.`` /
    ``ssl_model.`` / ``model.`` / ``w2v_encoder.w2v_model.`` prefixes; pre-training heads
    (quantizer, project_q, final_proj, mask_emb) are ignored; a checkpoint with more encoder layers
    than the trunk keeps the FIRST ``len(trunk.encoder.layers)`` (models/fe.py:72-74)."""
    try:
        ck = torch.load(path, map_location="cpu", weights_only=True)
    except Exception:
        ck = torch.load(path, map_location="cpu", weights_only=False, pickle_module=_StubPickle)
    sd = ck.get("model", ck) if isinstance(ck, dict) else ck
    if not isinstance(sd, dict):
        raise KeyError(f"{path}: no state_dict found in the checkpoint")
    out = {}
    for k, v in sd.items():
        if not torch.is_tensor(v):
            continue
        for pre in ("module.", "w2v_encoder.", "w2v_model.", "ssl_model.", "model."):
            if k.startswith(pre):
                k = k[len(pre):]
        if k.startswith(_UNUSED_SSL):
            continue
        out[k] = v
    own = trunk.state_dict()
    missing = [k for k in own if k not in out]
    if missing:
        raise KeyError(f"{path}: checkpoint lacks {len(missing)} trunk tensors, e.g. {missing[:3]}")
    bad = [k for k in own if tuple(out[k].shape) != tuple(own[k].shape)]
    if bad:
        raise ValueError(f"{path}: shape mismatch for {bad[:3]}")
    trunk.load_state_dict({k: out[k].to(torch.float32) for k in own}, strict=True)


# ---- engine routing ---------------------------------------------------------------------
class AfxModule(nn.Module):
    """nn.Module whose eval-mode forward runs on the native MI355X engine.  The
    engine's packed weights are refreshed whenever the module's parameters change
    (load_state_dict, .to(), in-place edits are caught by the version counters)."""

    afx_arch = "ssl"

    def _afx_cfg(self):
        return {}

    def _afx_signature(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters()) + \
            tuple((b.data_ptr(), b._version) for b in self.buffers())

    def _afx_engine(self):
        eng = self.__dict__.get("_afx_eng")
        n_layers = len(self._afx_trunk().encoder.layers)
        dtype = self.__dict__.get("afx_dtype", None) or DEFAULT_DTYPE
        dev = next(self.parameters()).device  # the engine lives where the module's parameters live (device=rank, main.py:48)
        if dev.type != "cuda":
            raise RuntimeError("the model is on the CPU: move it to a GPU first (the native path has no CPU fallback)")
        key = (n_layers, dtype, str(dev), tuple(sorted(self._afx_cfg().items())))
        if eng is None or self.__dict__.get("_afx_key") != key:
            eng = Engine(self.afx_arch, n_layers=n_layers, dtype=dtype, device=dev, **self._afx_cfg())
            self.__dict__["_afx_eng"], self.__dict__["_afx_key"], self.__dict__["_afx_sig"] = eng, key, None
        sig = self._afx_signature()
        if self.__dict__.get("_afx_sig") != sig:
            eng.load_state_dict(self.state_dict())
            self.__dict__["_afx_sig"] = sig
        return eng

    def _afx_check(self, x):
        if self.training:
            raise RuntimeError("the MI355X-native path is inference-only: call model.eval() first "
                               "(main.py:202, trainer.py:86)")
        if not x.is_cuda:
            raise RuntimeError("input must be on the GPU: the native path has no CPU fallback")

    def forward_ragged(self, clips):
        """Score clips of different lengths in one native call, each exactly as ``self(clip[None])`` would
        (key-padding masks; the reference reaches un-cropped clips only with batch size 1)."""
        if self.training:
            raise RuntimeError("the MI355X-native path is inference-only: call model.eval() first")
        eng = self._afx_engine()
        return eng.ssl_ragged(clips) if self.afx_arch == "ssl" else eng.forward_ragged(clips)

    def forward_overlapped(self, x):
        """``self(x)`` issued so that two batches are in flight: either the back-end of this batch on a side stream under the NEXT
        call's trunk, or whole forwards of consecutive calls on alternating streams (Engine.forward_overlapped / forward_lanes;
        ``overlap_pays`` times both against one stream and picks); the logits must not be read on the current stream before
        ``join_overlapped()``.  For scoring loops that collect scores and read them once at the end
        (afx.harness.produce_evaluation_file); ``self(x)`` itself stays one stream.  Same bits in every form."""
        x = x.squeeze(-1) if x.ndim == 3 else x
        self._afx_check(x)
        return self._afx_engine().forward_overlapped(x)

    def overlap_pays(self, x):
        """Engine.overlap_pays on this model's engine: times one stream and the two two-stream forms on ``x``, makes
        ``forward_overlapped`` issue the faster two-stream form, and says whether that beats one stream in this process."""
        x = x.squeeze(-1) if x.ndim == 3 else x
        self._afx_check(x)
        return self._afx_engine().overlap_pays(x)

    def join_overlapped(self):
        eng = self.__dict__.get("_afx_eng")
        if eng is not None:
            eng.join()

    def check_finite(self):
        """Engine.check_finite: raises if a forward since the last check overflowed its operand precision (non-finite
        features / logits) -- the scoring loops call it before they write a score file."""
        eng = self.__dict__.get("_afx_eng")
        if eng is not None:
            eng.check_finite()

    def set_precision(self, dtype):
        """'fp16' (default) or 'bf16' matrix-core operands; 'fp16x3' (split precision, fp32-accurate, ~1/3 of the rate) or
        'fp32' (exact mode, 1/16) where every score must hold the tolerance whatever the checkpoint's top-k gaps."""
        self.__dict__["afx_dtype"] = dtype
        return self


def resolve_device(device):
    """``device`` as the reference passes it: an int rank (main.py:48) or 'cuda'/'cpu'."""
    if isinstance(device, int):
        return torch.device("cuda", device) if torch.cuda.is_available() else torch.device("cpu")
    return torch.device(device)


def ssl_checkpoint_or_synthetic(trunk, path):
    if path and os.path.exists(path):
        load_ssl_checkpoint(trunk, path)
        return True
    return False
