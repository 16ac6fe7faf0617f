"""Oracle: AASIST graph-attention back-end, fp32 torch CPU (TEST INFRASTRUCTURE).

Functional restatement of ``models/xlsr_aasist.py:86-177`` (XLSR_AASIST.forward
after the SSL trunk) and the four modules of ``models/aasist_modules.py``.
Weights come as a dict with the reference's own state_dict key names (``LL.*``,
``first_bn.*``, ``encoder.{i}.0.*``, ``attention.{0,2,3}.*``, ``pos_S``,
``master1/2``, ``GAT_layer_{S,T}.*``, ``HtrgGAT_layer_ST{11,12,21,22}.*``,
``pool_*.proj.*``, ``out_layer.*``).  Eval mode only: every dropout is the
identity, every BatchNorm uses its running statistics.

The reference's numeric quirks are reproduced on purpose:
  Q1  ``out_S1 = out_S1 + 1``            (models/xlsr_aasist.py:138)
  Q2  ``Residual_block``: bn1+selu result is discarded, conv1 sees ``x``
      (models/aasist_modules.py:376-383)
  Q3  the un-expanded (1,1,64) master parameter is what the first heterogeneous
      layer of each branch receives (models/xlsr_aasist.py:125-130,142-143)
Pinned against the reference's own code: tests/golden/aasist_*.npz.
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def _bn(sd, p, x, dim):
    """Eval BatchNorm over channel axis ``dim`` with running stats."""
    shape = [1] * x.ndim
    shape[dim] = -1
    m = sd[p + "running_mean"].view(shape)
    v = sd[p + "running_var"].view(shape)
    return (x - m) / torch.sqrt(v + BN_EPS) * sd[p + "weight"].view(shape) + sd[p + "bias"].view(shape)


def residual_block(sd, p, x, first):
    """models/aasist_modules.py:340-397.  conv1 runs on x (Q2); bn1 unused."""
    out = F.conv2d(x, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=(1, 1))
    out = F.selu(_bn(sd, p + "bn2.", out, 1))
    out = F.conv2d(out, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=(0, 1))
    if (p + "conv_downsample.weight") in sd:
        x = F.conv2d(x, sd[p + "conv_downsample.weight"], sd[p + "conv_downsample.bias"], padding=(0, 1))
    return out + x


def graph_attention(sd, p, x, temp):
    """GraphAttentionLayer, models/aasist_modules.py:17-110."""
    pair = x.unsqueeze(2) * x.unsqueeze(1)  # (B,N,N,D): [b,i,j] = x_i * x_j
    a = torch.tanh(F.linear(pair, sd[p + "att_proj.weight"], sd[p + "att_proj.bias"]))
    a = torch.matmul(a, sd[p + "att_weight"]) / temp  # (B,N,N,1)
    a = torch.softmax(a, dim=-2).squeeze(-1)
    y = F.linear(torch.matmul(a, x), sd[p + "proj_with_att.weight"], sd[p + "proj_with_att.bias"]) \
        + F.linear(x, sd[p + "proj_without_att.weight"], sd[p + "proj_without_att.bias"])
    return F.selu(_bn(sd, p + "bn.", y, 2))


def graph_pool(sd, p, h, k, taps=None):
    """GraphPool, models/aasist_modules.py:296-338 (descending top-k order).  With ``taps``: the kept indices and,
    per utterance, the smallest score gap that decides membership or order of the kept nodes (the discontinuity of
    the reference model: a perturbation larger than that gap changes which nodes are paired downstream)."""
    s = torch.sigmoid(F.linear(h, sd[p + "proj.weight"], sd[p + "proj.bias"]))
    n = max(int(h.shape[1] * k), 1)
    _, idx = torch.topk(s, n, dim=1)
    if taps is not None:
        v, _ = torch.sort(s.squeeze(-1), dim=1, descending=True)
        gap = (v[:, :n] - v[:, 1:n + 1]).min(dim=1)[0] if v.shape[1] > n else (v[:, :n - 1] - v[:, 1:n]).min(dim=1)[0]
        taps.setdefault("pool_idx", {})[p] = idx.squeeze(-1)
        taps.setdefault("pool_margin", {})[p] = gap
    return torch.gather(h * s, 1, idx.expand(-1, -1, h.shape[2]))


def htrg_graph_attention(sd, p, x1, x2, master, temp):
    """HtrgGraphAttentionLayer, models/aasist_modules.py:112-294."""
    n1, n2 = x1.shape[1], x2.shape[1]
    x1 = F.linear(x1, sd[p + "proj_type1.weight"], sd[p + "proj_type1.bias"])
    x2 = F.linear(x2, sd[p + "proj_type2.weight"], sd[p + "proj_type2.bias"])
    x = torch.cat([x1, x2], dim=1)
    if master is None:
        master = x.mean(dim=1, keepdim=True)
    # node attention map with block-wise weight vectors (:239-267)
    pair = x.unsqueeze(2) * x.unsqueeze(1)
    a = torch.tanh(F.linear(pair, sd[p + "att_proj.weight"], sd[p + "att_proj.bias"]))
    board = torch.zeros_like(a[..., :1])
    board[:, :n1, :n1] = torch.matmul(a[:, :n1, :n1], sd[p + "att_weight11"])
    board[:, n1:, n1:] = torch.matmul(a[:, n1:, n1:], sd[p + "att_weight22"])
    board[:, :n1, n1:] = torch.matmul(a[:, :n1, n1:], sd[p + "att_weight12"])
    board[:, n1:, :n1] = torch.matmul(a[:, n1:, :n1], sd[p + "att_weight12"])
    att = torch.softmax(board / temp, dim=-2).squeeze(-1)
    # master update uses x before the node update (:201-206,222-237,275-281)
    am = torch.tanh(F.linear(x * master, sd[p + "att_projM.weight"], sd[p + "att_projM.bias"]))
    am = torch.softmax(torch.matmul(am, sd[p + "att_weightM"]) / temp, dim=-2)  # (B,N,1)
    new_master = F.linear(torch.matmul(am.squeeze(-1).unsqueeze(1), x),
                          sd[p + "proj_with_attM.weight"], sd[p + "proj_with_attM.bias"]) \
        + F.linear(master, sd[p + "proj_without_attM.weight"], sd[p + "proj_without_attM.bias"])
    y = F.linear(torch.matmul(att, x), sd[p + "proj_with_att.weight"], sd[p + "proj_with_att.bias"]) \
        + F.linear(x, sd[p + "proj_without_att.weight"], sd[p + "proj_without_att.bias"])
    y = F.selu(_bn(sd, p + "bn.", y, 2))
    return y[:, :n1], y[:, n1:], new_master


def aasist_front(sd, feats):
    """models/xlsr_aasist.py:89-118: (B,T,1024) -> e_S (B,42,64), e_T (B,T//3,64)."""
    x = F.linear(feats, sd["LL.weight"], sd["LL.bias"])
    x = x.transpose(1, 2).unsqueeze(1)
    x = F.max_pool2d(x, (3, 3))
    x = F.selu(_bn(sd, "first_bn.", x, 1))
    for i in range(6):
        x = residual_block(sd, f"encoder.{i}.0.", x, first=(i == 0))
    x = F.selu(_bn(sd, "first_bn1.", x, 1))
    w = F.conv2d(x, sd["attention.0.weight"], sd["attention.0.bias"])
    w = _bn(sd, "attention.2.", F.selu(w), 1)
    w = F.conv2d(w, sd["attention.3.weight"], sd["attention.3.bias"])
    e_S = (x * torch.softmax(w, dim=-1)).sum(dim=-1).transpose(1, 2) + sd["pos_S"]
    e_T = (x * torch.softmax(w, dim=-2)).sum(dim=-2).transpose(1, 2)
    return e_S, e_T


def aasist_graph(sd, e_S, e_T, taps=None):
    """models/xlsr_aasist.py:111-177."""
    out_S = graph_pool(sd, "pool_S.", graph_attention(sd, "GAT_layer_S.", e_S, 2.0), 0.5, taps)
    out_T = graph_pool(sd, "pool_T.", graph_attention(sd, "GAT_layer_T.", e_T, 2.0), 0.5, taps)
    if taps is not None:
        taps["out_S"], taps["out_T"] = out_S, out_T
    # branch 1 (Q3: raw parameter as master)
    T1, S1, m1 = htrg_graph_attention(sd, "HtrgGAT_layer_ST11.", out_T, out_S, sd["master1"], 100.0)
    S1 = graph_pool(sd, "pool_hS1.", S1, 0.5, taps)
    T1 = graph_pool(sd, "pool_hT1.", T1, 0.5, taps)
    Ta, Sa, ma = htrg_graph_attention(sd, "HtrgGAT_layer_ST12.", T1, S1, m1, 100.0)
    T1 = T1 + Ta
    S1 = S1 + 1  # Q1
    m1 = m1 + ma
    # branch 2
    T2, S2, m2 = htrg_graph_attention(sd, "HtrgGAT_layer_ST21.", out_T, out_S, sd["master2"], 100.0)
    S2 = graph_pool(sd, "pool_hS2.", S2, 0.5, taps)
    T2 = graph_pool(sd, "pool_hT2.", T2, 0.5, taps)
    Ta, Sa, ma = htrg_graph_attention(sd, "HtrgGAT_layer_ST22.", T2, S2, m2, 100.0)
    T2 = T2 + Ta
    S2 = S2 + Sa
    m2 = m2 + ma
    oT = torch.max(T1, T2)
    oS = torch.max(S1, S2)
    m = torch.max(m1, m2)
    hid = torch.cat([oT.abs().max(dim=1)[0], oT.mean(dim=1),
                     oS.abs().max(dim=1)[0], oS.mean(dim=1), m.squeeze(1)], dim=1)
    if taps is not None:
        taps["hidden"] = hid
    return F.linear(hid, sd["out_layer.weight"], sd["out_layer.bias"])


def aasist_backend(sd, feats, taps=None):
    """(B,T,1024) SSL features -> (B,2) logits."""
    e_S, e_T = aasist_front(sd, feats)
    if taps is not None:
        taps["e_S"], taps["e_T"] = e_S, e_T
    return aasist_graph(sd, e_S, e_T, taps)
