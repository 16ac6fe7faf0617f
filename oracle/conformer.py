"""Oracle: Conformer student head, fp32 torch CPU (TEST INFRASTRUCTURE).

PARTIALLY PINNED at the block level: ``conformer.ConformerBlock`` is the PyPI
package ``conformer`` (lucidrains/conformer; third-party, unpinned, absent from
this image) and the reference holds no test/golden vector for it.  The block is
checked against two in-container ``transformers`` implementations of the same
algorithms (``tests/golden/conformer_block.npz``, ``conformer_attn_shaw.npz``);
the (query - key) index convention of the relative-position table and the even-
kernel padding remain PARITY UNPINNED.  The block
below restates that package's published algorithm (SURVEY.md 8(a) row 12 and
appendix A.3) and is anchored on the reference call site
``models/conformer_baseline.py:16-18`` (``ConformerBlock(dim=emb, dim_head=
emb/heads, heads, ff_mult=4, conv_expansion_factor=2, conv_kernel_size)``) and
on the state_dict key names a reference checkpoint carries:

  ff1.fn.norm / ff1.fn.fn.net.{0,3}   half-step feed-forward (Swish)
  attn.norm / attn.fn.{to_q,to_kv,to_out,rel_pos_emb}
  conv.net.{0 LN, 2 pointwise, 4.conv depthwise, 5 BatchNorm1d, 7 pointwise}
  ff2.fn.* , post_norm

      x = x + 0.5*FF1(LN(x)); x = x + MHSA(LN(x)); x = x + Conv(x);
      x = x + 0.5*FF2(LN(x)); x = LN(x)

MHSA uses Shaw relative positions: ``logits[i,j] = (q_i.k_j + q_i.E[clamp(i-j,
+-512)+512]) * dh**-0.5``.  Depthwise conv uses "same" padding
``(k//2, k//2 - (k+1)%2)``.

The surrounding head -- ``MyConformer.forward`` (models/conformer_baseline.py:
22-29) and ``Model.forward`` (:54-64) -- is restated from the reference file
itself.
"""
import torch
import torch.nn.functional as F

LN_EPS = 1e-5
BN_EPS = 1e-5
MAX_POS = 512


def _id(t):
    return t


def _swish(x):
    return x * torch.sigmoid(x)


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + "weight"], sd[p + "bias"], LN_EPS)


def feed_forward(sd, p, x, q=_id):
    h = _ln(sd, p + "fn.norm.", x)
    h = _swish(F.linear(q(h), q(sd[p + "fn.fn.net.0.weight"]), sd[p + "fn.fn.net.0.bias"]))
    return F.linear(q(h), q(sd[p + "fn.fn.net.3.weight"]), sd[p + "fn.fn.net.3.bias"])


def attention(sd, p, x, heads, q=_id):
    B, N, D = x.shape
    h = q(_ln(sd, p + "norm.", x))
    wq = sd[p + "fn.to_q.weight"]
    inner = wq.shape[0]
    dh = inner // heads
    qq = F.linear(h, q(wq))
    kv = F.linear(h, q(sd[p + "fn.to_kv.weight"]))
    kk, vv = kv[..., :inner], kv[..., inner:]
    qq = qq.view(B, N, heads, dh).transpose(1, 2)
    kk = kk.view(B, N, heads, dh).transpose(1, 2)
    vv = vv.view(B, N, heads, dh).transpose(1, 2)
    scale = dh ** -0.5
    dots = torch.einsum("bhid,bhjd->bhij", q(qq), q(kk)) * scale
    seq = torch.arange(N)
    dist = (seq[:, None] - seq[None, :]).clamp(-MAX_POS, MAX_POS) + MAX_POS
    rel = sd[p + "fn.rel_pos_emb.weight"][dist]  # (N,N,dh)
    dots = dots + torch.einsum("bhnd,nrd->bhnr", q(qq), q(rel)) * scale
    att = torch.softmax(dots, dim=-1)
    out = torch.einsum("bhij,bhjd->bhid", q(att), q(vv)).transpose(1, 2).reshape(B, N, inner)
    return F.linear(q(out), q(sd[p + "fn.to_out.weight"]), sd[p + "fn.to_out.bias"])


def conv_module(sd, p, x, q=_id):
    h = _ln(sd, p + "net.0.", x).transpose(1, 2)  # (B,C,N)
    h = F.conv1d(q(h), q(sd[p + "net.2.weight"]), sd[p + "net.2.bias"])
    a, g = h.chunk(2, dim=1)
    h = a * torch.sigmoid(g)
    k = sd[p + "net.4.conv.weight"].shape[-1]
    pad = (k // 2, k // 2 - (k + 1) % 2)
    h = F.conv1d(F.pad(h, pad), sd[p + "net.4.conv.weight"], sd[p + "net.4.conv.bias"], groups=h.shape[1])
    m = sd[p + "net.5.running_mean"][None, :, None]
    v = sd[p + "net.5.running_var"][None, :, None]
    h = (h - m) / torch.sqrt(v + BN_EPS) * sd[p + "net.5.weight"][None, :, None] + sd[p + "net.5.bias"][None, :, None]
    h = _swish(h)
    h = F.conv1d(q(h), q(sd[p + "net.7.weight"]), sd[p + "net.7.bias"])
    return h.transpose(1, 2)


def conformer_block(sd, p, x, heads, q=None):
    q = q or _id
    x = x + 0.5 * feed_forward(sd, p + "ff1.", x, q)
    x = x + attention(sd, p + "attn.", x, heads, q)
    x = x + conv_module(sd, p + "conv.", x, q)
    x = x + 0.5 * feed_forward(sd, p + "ff2.", x, q)
    return _ln(sd, p + "post_norm.", x)


def conformer_head(sd, feats, heads=4, q=None, taps=None):
    """models/conformer_baseline.py:54-64 + :22-29.  (B,T,1024) -> (B,2).

    LL -> BatchNorm2d(1) (eval) -> SELU -> prepend class token -> N blocks ->
    token 0 -> fc5.
    """
    qf = q or _id
    x = F.linear(qf(feats), qf(sd["LL.weight"]), sd["LL.bias"])
    x = (x - sd["first_bn.running_mean"]) / torch.sqrt(sd["first_bn.running_var"] + BN_EPS) \
        * sd["first_bn.weight"] + sd["first_bn.bias"]
    x = F.selu(x)
    return my_conformer(sd, x, heads=heads, q=q, taps=taps)[0]


def my_conformer(sd, x, heads=4, q=None, taps=None, prefix="conformer."):
    """``MyConformer.forward`` (models/conformer_baseline.py:22-29): x (B,T,emb) -> (logits (B,2), embedding (B,emb)).
    Class token prepended (:23-24), the cloned blocks (:25-26), token 0 (:27), fc5 (:28)."""
    tok = sd[prefix + "class_token"].unsqueeze(0).expand(x.shape[0], -1, -1)
    x = torch.cat([tok, x], dim=1)
    if taps is not None:
        taps["tokens"] = x
    n = 0
    while f"{prefix}encoder_blocks.{n}.post_norm.weight" in sd:
        x = conformer_block(sd, f"{prefix}encoder_blocks.{n}.", x, heads, q)
        if taps is not None:
            taps[f"block{n}"] = x
        n += 1
    emb = x[:, 0, :]
    return F.linear(emb, sd[prefix + "fc5.weight"], sd[prefix + "fc5.bias"]), emb
