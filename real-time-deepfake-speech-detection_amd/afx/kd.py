"""Intermediate activations by module path, for the knowledge-distillation harness.

The reference's KDTrainer (trainer.py:156-195) registers forward hooks through torchdistill's
``ForwardHookManager(device).add_hook(model, module_path, requires_input=..., requires_output=...)`` and, after a
forward, reads ``pop_io_dict()`` -> ``{module_path: {"input": ..., "output": ...}}`` (trainer.py:263-270).  The drop-in
models here run their forward as ONE native call, so sub-module hooks never fire; this manager keeps that interface on
top of the engine's taps: the same paths, the same dictionary, filled from fp32 copies of the intermediates the native
forward keeps when taps are on.  Inference (eval) side only -- the teacher of the reference's KD loop.

Supported paths (B = batch, T = SSL frames, E = Conformer width):
  ssl_model, ssl_model.model                      output (B,T,1024)   final encoder LayerNorm = extract_feat()
  ssl_model.model.encoder.layers.<i>              input / output (B,T,1024): the residual stream around layer i
                                                  (batch-first; fairseq's own hook would see (T,B,C) and a tuple)
  LL                                              input (B,T,1024)
  conformer.encoder_blocks.<i>                    input / output (B,T+1,E)
  GAT_layer_S / GAT_layer_T                       input (B,42,64) / (B,T//3,64)
  out_layer                                       input (B,160), output (B,2)
Anything else raises at add_hook time (a silent missing key would only surface as a KeyError inside a loss)."""
import re

import torch


class ForwardHookManager:
    def __init__(self, target_device=None):
        self.target_device = torch.device(target_device) if target_device is not None else None
        self.io_dict = {}
        self._wanted = {}   # id(model) -> {path: (requires_input, requires_output)}
        self._models = {}   # id(model) -> model (to turn its engine's taps off again)
        self._handles = []

    # ---- path -> (input tap, output tap, shape fixers) ---------------------------------------
    @staticmethod
    def _taps_for(model, path):
        n_layers = len(model._afx_trunk().encoder.layers)
        m = re.fullmatch(r"ssl_model\.model\.encoder\.layers\.(\d+)", path)
        if m:
            i = int(m.group(1))
            if i >= n_layers:
                raise ValueError(f"{path}: the trunk has {n_layers} layers")
            return ("pos" if i == 0 else f"layer{i - 1}"), f"layer{i}", 1024
        if path in ("ssl_model", "ssl_model.model"):
            return None, "ssl", 1024
        if path == "LL":
            return "ssl", None, 1024
        m = re.fullmatch(r"conformer\.encoder_blocks\.(\d+)", path)
        if m and model.afx_arch == "conformer":
            i = int(m.group(1))
            return ("tokens" if i == 0 else f"block{i - 1}"), f"block{i}", model._afx_cfg()["conf_emb"]
        if model.afx_arch == "xlsr_aasist":
            if path == "GAT_layer_S":
                return "e_S", None, 64
            if path == "GAT_layer_T":
                return "e_T", None, 64
            if path == "out_layer":
                return "hidden", "__logits__", 160
        raise ValueError(f"no native tap behind module path '{path}' (see afx.kd for the supported ones)")

    def add_hook(self, model, module_path, requires_input=True, requires_output=True, **_kw):
        model.get_submodule(module_path)  # AttributeError for a path the model does not have, like torchdistill
        self._taps_for(model, module_path)  # and ValueError for one the native forward does not expose
        first = id(model) not in self._wanted
        self._wanted.setdefault(id(model), {})[module_path] = (bool(requires_input), bool(requires_output))
        if first:
            self._models[id(model)] = model
            self._handles.append(model.register_forward_pre_hook(lambda mod, args: mod._afx_engine().enable_taps(True)))
            self._handles.append(model.register_forward_hook(self._collect))

    def _collect(self, model, args, output):
        eng = model._afx_engine()
        B = output.shape[0]
        for path, (want_in, want_out) in self._wanted[id(model)].items():
            tin, tout, width = self._taps_for(model, path)
            rec = {}
            if want_in and tin is not None:
                rec["input"] = self._move(eng.tap(tin).reshape(B, -1, width) if tin != "hidden" else eng.tap(tin).reshape(B, width))
            if want_out and tout is not None:
                rec["output"] = self._move(output if tout == "__logits__" else eng.tap(tout).reshape(B, -1, width))
            self.io_dict[path] = rec

    def _move(self, t):
        return t.to(self.target_device) if self.target_device is not None else t

    def pop_io_dict(self):
        out, self.io_dict = self.io_dict, {}
        return out

    def clear(self):
        """Remove the hooks AND switch the engines' taps off again: with taps on every forward keeps fp32 copies of all
        intermediates (device allocations, D2D copies, half -> fp32 kernels -- none of it capturable in a hipGraph)."""
        for h in self._handles:
            h.remove()
        for model in self._models.values():
            eng = model.__dict__.get("_afx_eng")
            if eng is not None:
                eng.enable_taps(False)
        self._handles, self._wanted, self._models, self.io_dict = [], {}, {}, {}
