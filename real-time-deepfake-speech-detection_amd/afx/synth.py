"""Deterministic synthetic weights and waveforms (no checkpoints/datasets exist
offline).  Every tensor is drawn from its own generator seeded with
``crc32(name)`` so the CPU oracle, the device path, every rank and every box see
bit-identical fp32 values.  Key names are the reference checkpoint's
(SURVEY.md section 5 / appendix A.2-A.3).

Normalisation parameters are deliberately *not* at their identity defaults
(gamma ~ 1+-0.1, beta ~ +-0.1, running stats perturbed) so that a kernel that
drops an affine term or a running statistic fails parity.
"""
import math
import zlib

import torch

SSL_PREFIX = "ssl_model.model."
CONV_LAYERS = [(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)] * 2


def _gen(name):
    g = torch.Generator(device="cpu")
    g.manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    return g


def _randn(name, shape, std=1.0, mean=0.0):
    return torch.randn(*shape, generator=_gen(name), dtype=torch.float32) * std + mean


def _rand(name, shape, lo=0.0, hi=1.0):
    return torch.rand(*shape, generator=_gen(name), dtype=torch.float32) * (hi - lo) + lo


def _linear(sd, name, out_f, in_f, std=None, bias=True):
    # default: the std of torch's own nn.Linear init, U(+-1/sqrt(fan_in))
    std = std if std is not None else 1.0 / math.sqrt(3.0 * in_f)
    sd[name + ".weight"] = _randn(name + ".weight", (out_f, in_f), std)
    if bias:
        sd[name + ".bias"] = _randn(name + ".bias", (out_f,), 0.02)


def _norm(sd, name, n):
    sd[name + ".weight"] = _randn(name + ".weight", (n,), 0.1, 1.0)
    sd[name + ".bias"] = _randn(name + ".bias", (n,), 0.1)


def _bn(sd, name, n):
    _norm(sd, name, n)
    sd[name + ".running_mean"] = _randn(name + ".running_mean", (n,), 0.1)
    sd[name + ".running_var"] = _rand(name + ".running_var", (n,), 0.5, 1.5)
    sd[name + ".num_batches_tracked"] = torch.tensor(100, dtype=torch.long)


def ssl_state_dict(n_layers=24, prefix=SSL_PREFIX, dim=1024, ffn=4096, conv_dim=512,
                   conv_layers=None, pos_k=128, pos_groups=16, extractor_mode="layer_norm"):
    """fairseq-named XLS-R trunk (layer_norm extractor mode, biased convs); ``extractor_mode="group_norm"`` gives the
    wav2vec2-base feature extractor instead (bias-free convs, GroupNorm affine of layer 0 under ``...0.2.*``)."""
    conv_layers = conv_layers or [(conv_dim, k, s) for (_, k, s) in CONV_LAYERS]
    sd = {}
    cin = 1
    for i, (c, k, s) in enumerate(conv_layers):
        n = f"{prefix}feature_extractor.conv_layers.{i}"
        sd[n + ".0.weight"] = _randn(n + ".0.weight", (c, cin, k), math.sqrt(2.0 / (cin * k)))
        if extractor_mode == "layer_norm":
            sd[n + ".0.bias"] = _randn(n + ".0.bias", (c,), 0.02)
            _norm(sd, n + ".2.1", c)
        elif i == 0:
            _norm(sd, n + ".2", c)
        cin = c
    _norm(sd, prefix + "layer_norm", cin)
    _linear(sd, prefix + "post_extract_proj", dim, cin)
    n = prefix + "encoder.pos_conv.0"
    v = _randn(n + ".weight_v", (dim, dim // pos_groups, pos_k), math.sqrt(4.0 / (pos_k * dim)))
    sd[n + ".weight_v"] = v
    sd[n + ".weight_g"] = v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt() * _randn(n + ".weight_g", (1, 1, pos_k), 0.1, 1.0)
    sd[n + ".bias"] = _randn(n + ".bias", (dim,), 0.02)
    for l in range(n_layers):
        p = f"{prefix}encoder.layers.{l}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            _linear(sd, p + "self_attn." + nm, dim, dim)
        _norm(sd, p + "self_attn_layer_norm", dim)
        _linear(sd, p + "fc1", ffn, dim)
        _linear(sd, p + "fc2", dim, ffn)
        _norm(sd, p + "final_layer_norm", dim)
    _norm(sd, prefix + "encoder.layer_norm", dim)
    return sd


def lively(head, scale=1.5):
    """The seeded head with its matrices scaled (tests/golden/make_golden.py recipe): graph nodes then differ from each
    other and the GraphPool scores spread over (0.05, 0.9) instead of sitting within 1e-6 of one another, where the
    top-k order of the reference model is rounding noise of whichever BLAS computed it."""
    out = {}
    for k, v in head.items():
        hit = (k.endswith(".weight") and v.ndim >= 2 and not k.startswith("LL")) or "att_weight" in k
        out[k] = v * scale if hit else v
    return out


def aasist_head_state_dict(ssl_dim=1024):
    """models/xlsr_aasist.py:23-84 parameter set (447 242 parameters)."""
    sd = {}
    _linear(sd, "LL", 128, ssl_dim)
    _bn(sd, "first_bn", 1)
    _bn(sd, "first_bn1", 64)
    filts = [[1, 32], [32, 32], [32, 64], [64, 64], [64, 64], [64, 64]]
    for i, (ci, co) in enumerate(filts):
        p = f"encoder.{i}.0."
        if i > 0:
            _bn(sd, p + "bn1", ci)  # present in checkpoints, never used (Q2)
        sd[p + "conv1.weight"] = _randn(p + "conv1.weight", (co, ci, 2, 3), math.sqrt(1.0 / (3 * ci * 6)))
        sd[p + "conv1.bias"] = _randn(p + "conv1.bias", (co,), 0.02)
        _bn(sd, p + "bn2", co)
        sd[p + "conv2.weight"] = _randn(p + "conv2.weight", (co, co, 2, 3), math.sqrt(1.0 / (3 * co * 6)))
        sd[p + "conv2.bias"] = _randn(p + "conv2.bias", (co,), 0.02)
        if ci != co:
            sd[p + "conv_downsample.weight"] = _randn(p + "conv_downsample.weight", (co, ci, 1, 3), math.sqrt(1.0 / (3 * ci * 3)))
            sd[p + "conv_downsample.bias"] = _randn(p + "conv_downsample.bias", (co,), 0.02)
    sd["attention.0.weight"] = _randn("attention.0.weight", (128, 64, 1, 1), 0.072)
    sd["attention.0.bias"] = _randn("attention.0.bias", (128,), 0.02)
    _bn(sd, "attention.2", 128)
    sd["attention.3.weight"] = _randn("attention.3.weight", (64, 128, 1, 1), 0.051)
    sd["attention.3.bias"] = _randn("attention.3.bias", (64,), 0.02)
    sd["pos_S"] = _randn("pos_S", (1, 42, 64))
    sd["master1"] = _randn("master1", (1, 1, 64))
    sd["master2"] = _randn("master2", (1, 1, 64))

    def gat(name, di, do):
        _linear(sd, name + ".att_proj", do, di)
        sd[name + ".att_weight"] = _randn(name + ".att_weight", (do, 1), math.sqrt(2.0 / (do + 1)))
        _linear(sd, name + ".proj_with_att", do, di)
        _linear(sd, name + ".proj_without_att", do, di)
        _bn(sd, name + ".bn", do)

    def hgat(name, di, do):
        _linear(sd, name + ".proj_type1", di, di)
        _linear(sd, name + ".proj_type2", di, di)
        _linear(sd, name + ".att_proj", do, di)
        _linear(sd, name + ".att_projM", do, di)
        for w in ("att_weight11", "att_weight22", "att_weight12", "att_weightM"):
            sd[f"{name}.{w}"] = _randn(f"{name}.{w}", (do, 1), math.sqrt(2.0 / (do + 1)))
        _linear(sd, name + ".proj_with_att", do, di)
        _linear(sd, name + ".proj_without_att", do, di)
        _linear(sd, name + ".proj_with_attM", do, di)
        _linear(sd, name + ".proj_without_attM", do, di)
        _bn(sd, name + ".bn", do)

    gat("GAT_layer_S", 64, 64)
    gat("GAT_layer_T", 64, 64)
    hgat("HtrgGAT_layer_ST11", 64, 32)
    hgat("HtrgGAT_layer_ST12", 32, 32)
    hgat("HtrgGAT_layer_ST21", 64, 32)
    hgat("HtrgGAT_layer_ST22", 32, 32)
    for nm, d in (("pool_S", 64), ("pool_T", 64), ("pool_hS1", 32), ("pool_hT1", 32),
                  ("pool_hS2", 32), ("pool_hT2", 32)):
        _linear(sd, nm + ".proj", 1, d)
    _linear(sd, "out_layer", 2, 160)
    return sd


def conformer_head_state_dict(emb_size=144, heads=4, kernel_size=31, n_encoders=4,
                              ff_mult=4, exp_fac=2, ssl_dim=1024, max_pos=512):
    """models/conformer_baseline.py:32-52 + lucidrains ConformerBlock keys (A.3)."""
    sd = {}
    dh = emb_size // heads
    inner = dh * heads
    ci = emb_size * exp_fac
    _linear(sd, "LL", emb_size, ssl_dim)
    _bn(sd, "first_bn", 1)
    sd["conformer.class_token"] = _rand("conformer.class_token", (1, emb_size))
    for b in range(n_encoders):
        p = f"conformer.encoder_blocks.{b}."
        for ff in ("ff1", "ff2"):
            _norm(sd, p + ff + ".fn.norm", emb_size)
            _linear(sd, p + ff + ".fn.fn.net.0", emb_size * ff_mult, emb_size)
            _linear(sd, p + ff + ".fn.fn.net.3", emb_size, emb_size * ff_mult)
        _norm(sd, p + "attn.norm", emb_size)
        _linear(sd, p + "attn.fn.to_q", inner, emb_size, bias=False)
        _linear(sd, p + "attn.fn.to_kv", inner * 2, emb_size, bias=False)
        _linear(sd, p + "attn.fn.to_out", emb_size, inner)
        sd[p + "attn.fn.rel_pos_emb.weight"] = _randn(p + "attn.fn.rel_pos_emb.weight", (2 * max_pos + 1, dh), 1.0)
        _norm(sd, p + "conv.net.0", emb_size)
        sd[p + "conv.net.2.weight"] = _randn(p + "conv.net.2.weight", (ci * 2, emb_size, 1), 1.0 / math.sqrt(3.0 * emb_size))
        sd[p + "conv.net.2.bias"] = _randn(p + "conv.net.2.bias", (ci * 2,), 0.02)
        sd[p + "conv.net.4.conv.weight"] = _randn(p + "conv.net.4.conv.weight", (ci, 1, kernel_size), 1.0 / math.sqrt(3.0 * kernel_size))
        sd[p + "conv.net.4.conv.bias"] = _randn(p + "conv.net.4.conv.bias", (ci,), 0.02)
        _bn(sd, p + "conv.net.5", ci)
        sd[p + "conv.net.7.weight"] = _randn(p + "conv.net.7.weight", (emb_size, ci, 1), 1.0 / math.sqrt(3.0 * ci))
        sd[p + "conv.net.7.bias"] = _randn(p + "conv.net.7.bias", (emb_size,), 0.02)
        _norm(sd, p + "post_norm", emb_size)
    _linear(sd, "conformer.fc5", 2, emb_size)
    return sd


def model_state_dict(model, n_layers=24, **kw):
    """Full reference-format state_dict for 'XLSR_AASIST' | 'ConformerModel' (and
    their My_* student variants: pass the truncated ``n_layers``)."""
    sd = ssl_state_dict(n_layers)
    if model in ("XLSR_AASIST", "My_XLSR_AASIST"):
        head = aasist_head_state_dict()
        sd.update(lively(head, kw["head_scale"]) if kw.get("head_scale") else head)
    elif model in ("ConformerModel", "MyConformerModel", "Model", "MyModel"):
        sd.update(conformer_head_state_dict(**kw))
    else:
        raise ValueError(f"Model {model} not found.")
    return sd


def with_outliers(sd, gain=30.0, n=4):
    """A copy of a seeded state_dict with XLS-R-style OUTLIER CHANNELS (trained wav2vec2 / XLS-R trunks carry a few channels
    whose activations are orders of magnitude above the rest; seeded weights at init scale do not): in every transformer
    layer ``n`` rows of ``fc1`` (weight and bias: those FFN hidden units), ``n`` rows of ``out_proj`` (those residual
    channels) and ``n`` gains of both LayerNorms are multiplied by ``gain``.  With gain 30 - 300 the FFN hidden and the
    residual stream reach 1e2 - 1e4 -- what the fp16 operand copies of a half-precision engine have to hold
    (tests/test_gpu_outliers.py).  Which rows: seeded per layer, the same on every box."""
    out = dict(sd)
    layers = sorted({k.split("encoder.layers.")[1].split(".")[0] for k in sd if "encoder.layers." in k}, key=int)
    for l in layers:
        p = f"{SSL_PREFIX}encoder.layers.{l}."
        g = _gen(p + "outliers")
        dim, ffn = sd[p + "fc2.weight"].shape
        hid = torch.randperm(ffn, generator=g)[:n]
        ch = torch.randperm(dim, generator=g)[:n]
        ln = torch.randperm(dim, generator=g)[:n]
        for k, idx in ((p + "fc1.weight", hid), (p + "fc1.bias", hid), (p + "self_attn.out_proj.weight", ch),
                       (p + "self_attn_layer_norm.weight", ln), (p + "final_layer_norm.weight", ln)):
            t = sd[k].clone()
            t[idx] = t[idx] * gain
            out[k] = t
    return out


def waveforms(batch, length=64000, batch_idx=0, scale=0.1):
    """x = 0.1*randn(B, L), seed 1234 + batch_idx."""
    g = torch.Generator(device="cpu")
    g.manual_seed(1234 + batch_idx)
    return torch.randn(batch, length, generator=g, dtype=torch.float32) * scale
