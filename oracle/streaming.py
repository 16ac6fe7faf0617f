"""Oracle of the KV-cached streaming mode (TEST INFRASTRUCTURE) -- explicitly NOT reference parity.

The reference has no streaming mode: its SSL trunk is bidirectional over the clip (``models/fe.py:17-21`` runs fairseq's
``Wav2Vec2Model`` on the whole waveform; full self-attention, a centred k = 128 positional conv).  BASELINE config 5 names
"250 ms chunks with cached SSL-encoder KV state"; an encoder with cached keys / values sees every frame ONCE, when its chunk
arrives, which is a different function.  SURVEY.md section 7 therefore scopes the mode as: parity target = the build's OWN
offline block-causal restatement.  This file is that restatement, in fp32 torch on the CPU, built from the same per-op
functions as the trunk oracle (``oracle/ssl_trunk.py``) so that the only thing that differs from the reference function is
the visibility rule:

  * conv feature extractor, feature LayerNorm, ``post_extract_proj``: frame-local, identical to the offline trunk.  The
    conv stack is causal-local, so the frames a stream prefix of L samples yields are the first ``conv_out_lengths(L)[-1]``
    frames of the whole stream's; chunk c = the frames that became computable with hop c's samples.
  * positional conv: frame t of chunk c sees projected frames [t - 64, end of chunk c]; beyond the chunk: zeros (exactly
    what the offline conv's zero padding is at the end of a clip).
  * transformer layer: queries = the chunk's frames; keys / values = the chunk and the 15 chunks before it (every frame's
    K / V computed once, from that frame's own block-causal state).
  * final LayerNorm; the back-end (``oracle/aasist.py`` / ``oracle/conformer.py``) on the window of the last <= 200 frames.
"""
import torch
import torch.nn.functional as F

from . import aasist, conformer, models, ssl_trunk

CTX_CHUNKS = 16   # chunks a query sees, its own included (16 x 250 ms = 4 s)
HEAD_WINDOW = 200  # feature frames the back-end scores
POS_LEFT = 64


def chunk_sizes(total_samples, hop):
    """Frames of conv layer 6 that each hop completes: n_frames(prefix after hop c) - n_frames(prefix before it)."""
    sizes, prev = [], 0
    for end in range(hop, total_samples + 1, hop):
        n = max(ssl_trunk.conv_out_lengths(end)[-1], 0)
        sizes.append(n - prev)
        prev = n
    return sizes


@torch.no_grad()
def block_causal_scores(sd, stream, hop, heads=16, conf_heads=4, windows=None):
    """sd: a whole model's state_dict (reference key names); stream (S, total) fp32.  Returns (list of (S, 2) logits, one per hop
    that completed at least one frame; the chunk sizes).  windows: an optional list that receives every hop's feature window."""
    ssl, head = models.split(sd)
    is_conformer = "conformer.class_token" in head
    S = stream.shape[0]
    feats = ssl_trunk.feature_extractor(ssl, stream.float())                      # (S, T, 512): every frame of the stream
    C = feats.shape[-1]
    u = F.linear(F.layer_norm(feats, (C,), ssl["layer_norm.weight"], ssl["layer_norm.bias"], ssl_trunk.LN_EPS),
                 ssl["post_extract_proj.weight"], ssl["post_extract_proj.bias"])  # projected frames (S, T, 1024)
    w_pos, b_pos = ssl_trunk.pos_conv_weight(ssl), ssl["encoder.pos_conv.0.bias"]
    n_layers = ssl_trunk.num_layers(ssl)
    D = u.shape[-1]
    dh = D // heads
    kcache = [[] for _ in range(n_layers)]   # per layer: list over chunks of (S, n_c, D) keys / values
    vcache = [[] for _ in range(n_layers)]
    window = torch.zeros(S, 0, D)
    out, sizes, t0 = [], chunk_sizes(stream.shape[1], hop), 0
    for n in sizes:
        if n <= 0:
            continue
        t1 = t0 + n
        # positional conv on [t1 - n - 64, t1): the offline conv over that slice, zero-padded on both sides, gives frame t its
        # left context back to t - 64 (the slice holds it) and NOTHING to the right of the chunk
        lo = max(t0 - POS_LEFT, 0)
        seg = u[:, lo:t1]
        if t0 - POS_LEFT < 0:  # before the stream started: zeros, like the left padding of a clip
            seg = torch.cat([torch.zeros(S, POS_LEFT - t0, D), seg], dim=1)
        y = F.conv1d(seg.transpose(1, 2), w_pos, b_pos, padding=w_pos.shape[-1] // 2, groups=16)
        if w_pos.shape[-1] % 2 == 0:
            y = y[:, :, :-1]
        x = (seg + F.gelu(y).transpose(1, 2))[:, -n:]
        for l in range(n_layers):
            p = f"encoder.layers.{l}."
            h = F.layer_norm(x, (D,), ssl[p + "self_attn_layer_norm.weight"], ssl[p + "self_attn_layer_norm.bias"], ssl_trunk.LN_EPS)
            q = F.linear(h, ssl[p + "self_attn.q_proj.weight"], ssl[p + "self_attn.q_proj.bias"]) * dh ** -0.5
            kcache[l].append(F.linear(h, ssl[p + "self_attn.k_proj.weight"], ssl[p + "self_attn.k_proj.bias"]))
            vcache[l].append(F.linear(h, ssl[p + "self_attn.v_proj.weight"], ssl[p + "self_attn.v_proj.bias"]))
            kcache[l], vcache[l] = kcache[l][-CTX_CHUNKS:], vcache[l][-CTX_CHUNKS:]
            kk, vv = torch.cat(kcache[l], dim=1), torch.cat(vcache[l], dim=1)
            T = kk.shape[1]
            qh = q.view(S, n, heads, dh).transpose(1, 2)
            kh = kk.view(S, T, heads, dh).transpose(1, 2)
            vh = vv.view(S, T, heads, dh).transpose(1, 2)
            o = (torch.softmax(qh @ kh.transpose(-1, -2), dim=-1) @ vh).transpose(1, 2).reshape(S, n, D)
            x = x + F.linear(o, ssl[p + "self_attn.out_proj.weight"], ssl[p + "self_attn.out_proj.bias"])
            h = F.layer_norm(x, (D,), ssl[p + "final_layer_norm.weight"], ssl[p + "final_layer_norm.bias"], ssl_trunk.LN_EPS)
            h = F.gelu(F.linear(h, ssl[p + "fc1.weight"], ssl[p + "fc1.bias"]))
            x = x + F.linear(h, ssl[p + "fc2.weight"], ssl[p + "fc2.bias"])
        f = F.layer_norm(x, (D,), ssl["encoder.layer_norm.weight"], ssl["encoder.layer_norm.bias"], ssl_trunk.LN_EPS)
        window = torch.cat([window, f], dim=1)[:, -HEAD_WINDOW:]
        if windows is not None:
            windows.append(window)
        out.append(conformer.conformer_head(head, window, heads=conf_heads) if is_conformer else aasist.aasist_backend(head, window))
        t0 = t1
    return out, sizes
