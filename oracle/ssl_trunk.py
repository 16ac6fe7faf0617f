"""Oracle: wav2vec2 / XLS-R SSL trunk, fp32 torch CPU (TEST INFRASTRUCTURE).

Restates what ``fairseq.models.wav2vec.wav2vec2.Wav2Vec2Model.forward(source,
mask=False, features_only=True)['x']`` computes -- the call the reference makes
at ``models/fe.py:17-21``, ``models/fe.py:95-99`` and ``models/models.py:23-26``.
fairseq itself is an unpinned, absent third-party dependency; the algorithm
below follows its published XLS-R-300M configuration:

  * conv feature extractor, ``extractor_mode=layer_norm``, ``conv_bias=True``:
    7 x [Conv1d -> LayerNorm over channels (fp32) -> erf-GELU]
  * LayerNorm(512) -> Linear 512->1024 (``post_extract_proj``)
  * positional conv: weight-normed Conv1d(k=128, pad=64, groups=16), drop the
    last frame (SamePad for even k), erf-GELU, added to the input
  * N x pre-LN transformer layer (``layer_norm_first=True``), then the final
    encoder LayerNorm.

Weights are passed as a dict with the *fairseq* key names, i.e.
what sits under ``ssl_model.model.`` in a reference checkpoint.

``q`` is an optional operand-rounding hook (``lambda t: t.bfloat16().float()``)
used only by the precision study in tests: it rounds the operands of every
GEMM-shaped op exactly where the device path feeds the matrix cores, keeping
fp32 accumulation.  ``q=None`` is the oracle proper.
"""
import math

import torch
import torch.nn.functional as F

CONV_LAYERS = [(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)] * 2
LN_EPS = 1e-5


def _id(t):
    return t


def conv_out_lengths(L, layers=CONV_LAYERS):
    """Frame counts after each conv layer (no padding): floor((L-k)/s)+1."""
    out = []
    for _, k, s in layers:
        L = (L - k) // s + 1
        out.append(L)
    return out


def feature_extractor(sd, wave, q=None, layers=CONV_LAYERS, mode="layer_norm"):
    """fairseq ConvFeatureExtractionModel.  wave (B,L) fp32 -> (B,T,C) fp32.

    mode="layer_norm": XLS-R (every layer conv+bias -> LN(C) -> GELU).
    mode="default":    wav2vec2-base (bias-free conv; GroupNorm(C,C) on layer 0
                        only, i.e. per-channel normalisation over time).
    """
    q = q or _id
    x = wave.unsqueeze(1)  # (B,1,L)
    for i, (_, k, s) in enumerate(layers):
        w = sd[f"feature_extractor.conv_layers.{i}.0.weight"]
        b = sd.get(f"feature_extractor.conv_layers.{i}.0.bias")
        x = F.conv1d(q(x), q(w), b, stride=s)
        if mode == "layer_norm":
            g = sd[f"feature_extractor.conv_layers.{i}.2.1.weight"]
            be = sd[f"feature_extractor.conv_layers.{i}.2.1.bias"]
            x = F.layer_norm(x.transpose(1, 2), (x.shape[1],), g, be, LN_EPS).transpose(1, 2)
        elif i == 0:
            g = sd["feature_extractor.conv_layers.0.2.weight"]
            be = sd["feature_extractor.conv_layers.0.2.bias"]
            x = F.group_norm(x, x.shape[1], g, be, LN_EPS)
        x = F.gelu(x)
    return x.transpose(1, 2)  # (B,T,C)


def pos_conv_weight(sd):
    """Effective weight of the weight-normed positional conv (dim=2):
    w = g * v / ||v||, the norm taken over dims (0,1) for every tap."""
    if "encoder.pos_conv.0.weight" in sd:
        return sd["encoder.pos_conv.0.weight"]
    g = sd["encoder.pos_conv.0.weight_g"]  # (1,1,K)
    v = sd["encoder.pos_conv.0.weight_v"]  # (C, C/groups, K)
    norm = v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
    return g * v / norm


def pos_conv(sd, x, q=None, groups=16):
    """x (B,T,C) -> x + gelu(samepad(conv(x)))."""
    q = q or _id
    w = pos_conv_weight(sd)
    k = w.shape[-1]
    y = F.conv1d(q(x.transpose(1, 2)), q(w), sd["encoder.pos_conv.0.bias"],
                 padding=k // 2, groups=groups)
    if k % 2 == 0:
        y = y[:, :, :-1]
    return x + F.gelu(y).transpose(1, 2)


def encoder_layer(sd, x, n, heads, q=None):
    """fairseq TransformerSentenceEncoderLayer, layer_norm_first=True."""
    q = q or _id
    p = f"encoder.layers.{n}."
    B, T, D = x.shape
    dh = D // heads
    h = F.layer_norm(x, (D,), sd[p + "self_attn_layer_norm.weight"],
                     sd[p + "self_attn_layer_norm.bias"], LN_EPS)
    hq = q(h)
    qq = F.linear(hq, q(sd[p + "self_attn.q_proj.weight"]), sd[p + "self_attn.q_proj.bias"]) * dh ** -0.5
    kk = F.linear(hq, q(sd[p + "self_attn.k_proj.weight"]), sd[p + "self_attn.k_proj.bias"])
    vv = F.linear(hq, q(sd[p + "self_attn.v_proj.weight"]), sd[p + "self_attn.v_proj.bias"])
    qq = qq.view(B, T, heads, dh).transpose(1, 2)
    kk = kk.view(B, T, heads, dh).transpose(1, 2)
    vv = vv.view(B, T, heads, dh).transpose(1, 2)
    att = torch.softmax(q(qq) @ q(kk).transpose(-1, -2), dim=-1)
    o = (q(att) @ q(vv)).transpose(1, 2).reshape(B, T, D)
    o = F.linear(q(o), q(sd[p + "self_attn.out_proj.weight"]), sd[p + "self_attn.out_proj.bias"])
    x = x + o
    h = F.layer_norm(x, (D,), sd[p + "final_layer_norm.weight"],
                     sd[p + "final_layer_norm.bias"], LN_EPS)
    h = F.gelu(F.linear(q(h), q(sd[p + "fc1.weight"]), sd[p + "fc1.bias"]))
    h = F.linear(q(h), q(sd[p + "fc2.weight"]), sd[p + "fc2.bias"])
    return x + h


def num_layers(sd):
    n = 0
    while f"encoder.layers.{n}.fc1.weight" in sd:
        n += 1
    return n


def ssl_forward(sd, wave, heads=16, q=None, layers=CONV_LAYERS, mode="layer_norm",
                groups=16, taps=None):
    """(B,L) or (B,L,1) fp32 -> (B,T,D) fp32; ``extract_feat`` of the reference
    (models/fe.py:17-21: 3-D inputs use channel 0)."""
    qf = q or _id
    if wave.ndim == 3:
        wave = wave[:, :, 0]
    feats = feature_extractor(sd, wave.float(), q, layers, mode)
    if taps is not None:
        taps["conv"] = feats
    C = feats.shape[-1]
    feats = F.layer_norm(feats, (C,), sd["layer_norm.weight"], sd["layer_norm.bias"], LN_EPS)
    x = F.linear(qf(feats), qf(sd["post_extract_proj.weight"]), sd["post_extract_proj.bias"])
    if taps is not None:
        taps["proj"] = x
    x = pos_conv(sd, x, q, groups)
    if taps is not None:
        taps["pos"] = x
    for n in range(num_layers(sd)):
        x = encoder_layer(sd, x, n, heads, q)
        if taps is not None:
            taps[f"layer{n}"] = x
    D = x.shape[-1]
    return F.layer_norm(x, (D,), sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"], LN_EPS)


def select_layers(sd, num_layers_keep=24, order="first", custom_order=None, total=24):
    """models/fe.py:55-90 (My_XLSR_FE): keep a subset of encoder layers and
    re-index them 0..N-1.  Same validation and ValueErrors as the reference."""
    if num_layers_keep < 1 or num_layers_keep > 24:
        raise ValueError("Number of layers must be at least 1 and at most 24.")
    if order == "last":
        idx = list(range(total))[-num_layers_keep:]
    elif order == "first":
        idx = list(range(total))[:num_layers_keep]
    elif order == "middle":
        start = (total - num_layers_keep) // 2  # models/fe.py:43-50
        idx = list(range(start, start + num_layers_keep))
    else:
        if custom_order is None:
            raise ValueError("Custom order must be provided as a list of integers (0-23).")
        if type(custom_order) != list:
            raise ValueError("Custom order must be a list of integers.")
        idx = list(custom_order)
    out = {k: v for k, v in sd.items() if not k.startswith("encoder.layers.")}
    for new, old in enumerate(idx):
        pre = f"encoder.layers.{old}."
        for k, v in sd.items():
            if k.startswith(pre):
                out[f"encoder.layers.{new}." + k[len(pre):]] = v
    return out
