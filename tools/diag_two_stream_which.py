"""Diagnostic: teacher, dtype fp16x3, two-stream step (back-end of batch i on the side stream under the trunk of batch i+1).
Where the two-stream logits differ from the one-stream logits: was it the TRUNK that produced other features, or the BACK-END
that produced other logits from the same features?  After the sequence has drained, the back-end is run again, alone, on each
slot's workspace (the features the two-stream trunk left there): equal to the two-stream logits -> the trunk moved; equal to
the one-stream logits -> the back-end moved."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import engine, synth  # noqa: E402
from afx._lib import check, lib, ptr  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "fp16x3"
sd = synth.model_state_dict("XLSR_AASIST", n_layers=2, head_scale=1.5)
eng = engine.Engine("xlsr_aasist", n_layers=2, dtype=dtype)
eng.load_state_dict(sd)
l = lib()
B, L, N = 5, 16000, 8
waves = [synth.waveforms(B, L, batch_idx=700 + i).cuda() for i in range(N)]
pristine = [w.cpu().clone() for w in waves]
want = [eng.forward(w).clone() for w in waves]
nbytes = l.afx_workspace_bytes(eng._h, B, L)
slots = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(N)]  # one workspace per batch: nothing is reused
outs = [torch.empty(B, 2, device="cuda") for _ in range(N)]
side = engine.side_stream(eng.device)
cur = torch.cuda.current_stream()
torch.cuda.synchronize()
stop = int(sys.argv[2]) if len(sys.argv) > 2 else 0
l.afx_debug_set(b"aasist_stop", stop)
print(f"back-end stops after stage {stop} (0 = whole back-end)", flush=True)
for rep in range(int(os.environ.get('DIAG_PASSES', '3'))):
    for i in range(N):
        check(l.afx_trunk_forward(eng._h, ptr(waves[i]), B, L, ptr(slots[i]), nbytes, C.c_void_p(cur.cuda_stream)))
        ev = torch.cuda.Event()
        ev.record(cur)
        side.wait_event(ev)
        check(l.afx_head_from_workspace(eng._h, B, L, ptr(outs[i]), ptr(slots[i]), nbytes, C.c_void_p(side.cuda_stream)))
    torch.cuda.synchronize()
    line = []
    for i in range(N):
        if stop == 0 and torch.equal(outs[i], want[i]):
            continue
        again = torch.empty(B, 2, device="cuda")
        l.afx_debug_set(b"aasist_stop", 0)
        check(l.afx_head_from_workspace(eng._h, B, L, ptr(again), ptr(slots[i]), nbytes, C.c_void_p(cur.cuda_stream)))
        torch.cuda.synchronize()
        l.afx_debug_set(b"aasist_stop", stop)
        # where in the workspace do the trunk's leftovers differ from a one-stream trunk of the same batch?  (carve order:
        # bufA | bufB | tmp32 | feats_h | x | xpad | hbuf | qkv | att | ff | ssl_f | ssl_h | back-end | pair-form scratch)
        ref_ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        ref_ws.copy_(slots[i])  # (same stale bytes wherever nothing writes)
        check(l.afx_trunk_forward(eng._h, ptr(waves[i]), B, L, ptr(ref_ws), nbytes, C.c_void_p(cur.cuda_stream)))
        torch.cuda.synchronize()
        neq = (ref_ws != slots[i]).nonzero().flatten()
        if not torch.equal(waves[i].cpu(), pristine[i]):
            print(f"   batch {i}: THE INPUT WAVEFORM ON THE GPU CHANGED: {(waves[i].cpu() != pristine[i]).sum().item()} samples", flush=True)
        if neq.numel():
            # the conv-layer-0 region of bufA beyond what later layers overwrite (fp32 rows of 512): which frames, how far apart
            n0 = B * ((L - 10) // 5 + 1) * 512
            a = slots[i][: n0 * 4].view(torch.float32).view(-1, 512)
            r = ref_ws[: n0 * 4].view(torch.float32).view(-1, 512)
            rows = (a != r).any(dim=1).nonzero().flatten()
            rows = rows[rows > B * 799 * 2]  # (beyond the rows conv layers 2 / 4 wrote over)
            if rows.numel():
                k = int(rows[0])
                # which of the two versions is conv layer 0 of THIS batch's frame -- and is the other one some OTHER frame's row?
                import torch.nn.functional as Fn
                P0 = "ssl_model.model.feature_extractor.conv_layers.0."
                cw, cb, lg, lb = (sd[P0 + n_].double() for n_ in ("0.weight", "0.bias", "2.1.weight", "2.1.bias"))
                allw = torch.stack(pristine).double()  # (N, B, L)
                y = Fn.conv1d(allw.reshape(-1, 1, L), cw, cb, stride=5).transpose(1, 2)  # (N*B, T0, 512)
                y = Fn.gelu(Fn.layer_norm(y, (512,), lg, lb, 1e-5)).float().reshape(-1, 512)
                T0 = (L - 10) // 5 + 1
                mine = (i * B) * T0 + k
                # what kind of error is the two-stream row?  (a) ONE of the frame's 10 input samples had another value (an LDS
                # staging error), (b) the LayerNorm statistics of the row were other numbers (a wave-reduction error)
                bq, fq = k // T0, k % T0
                xwin = pristine[i][bq, 5 * fq:5 * fq + 10].double()
                vpre = cw[:, 0, :] @ xwin + cb  # (512,) conv output before the LayerNorm
                bad_row = a[k].cpu().double()
                best = None
                for jj in range(10):
                    ds = torch.linspace(-0.6, 0.6, 24001, dtype=torch.float64)
                    vv = vpre[None] + ds[:, None] * cw[None, :, 0, jj]
                    yy = Fn.gelu(Fn.layer_norm(vv, (512,), lg, lb, 1e-5))
                    e = (yy - bad_row[None]).abs().max(dim=1)[0]
                    m_ = int(e.argmin())
                    if best is None or float(e[m_]) < best[0]:
                        best = (float(e[m_]), jj, float(ds[m_]), float(xwin[jj]))
                dch = (a[k].cpu() - r[k].cpu()).abs()
                top = torch.topk(dch, 12)
                print(f"      largest channel errors: {[(int(c_), int(c_) // 8, round(float(v_), 4)) for v_, c_ in zip(top.values, top.indices)]} (channel, lane = channel // 8, |d|); median |d| {float(dch.median()):.2e}; frame % 64 = {fq % 64}, wave = {fq % 4}", flush=True)
                print(f"      (a) best single-sample explanation: tap {best[1]} with sample {best[3]:+.5f} changed by {best[2]:+.5f} leaves max |d| {best[0]:.2e}", flush=True)
                mu, sg = vpre.mean(), vpre.var(unbiased=False).add(1e-5).sqrt()
                bestb = None
                for dm in torch.linspace(-0.2, 0.2, 401, dtype=torch.float64):
                    sc_ = torch.linspace(0.8, 1.2, 801, dtype=torch.float64)
                    z = ((vpre[None] - mu - dm) / sg) * sc_[:, None]
                    yy = Fn.gelu(z * lg[None] + lb[None])
                    e = (yy - bad_row[None]).abs().max(dim=1)[0]
                    m_ = int(e.argmin())
                    if bestb is None or float(e[m_]) < bestb[0]:
                        bestb = (float(e[m_]), float(dm), float(sc_[m_]))
                print(f"      (b) best wrong-statistics explanation: mean off by {bestb[1]:+.4f}, 1/std scaled by {bestb[2]:.4f} leaves max |d| {bestb[0]:.2e}", flush=True)
                for name, row in (("two-stream", a[k].cpu()), ("one-stream", r[k].cpu())):
                    d = (y - row[None]).abs().max(dim=1)[0]
                    j = int(d.argmin())
                    print(f"      {name} row: |d| to this frame's exact value {float(d[mine]):.2e}; nearest exact row over all batches: batch {j // (B * T0)} utt {(j // T0) % B} frame {j % T0} at {float(d[j]):.2e}"
                          f" (this is batch {i} utt {k // T0} frame {k % T0})", flush=True)
                print(f"   batch {i}: conv-layer-0 rows that differ: {rows.tolist()[:8]}; row {k}: two-stream {a[k, :4].tolist()} one-stream {r[k, :4].tolist()} max |d| {float((a[k] - r[k]).abs().max()):.3e}", flush=True)
            idx = neq.cpu()
            cuts = [0] + (torch.nonzero(idx[1:] - idx[:-1] > (1 << 16)).flatten() + 1).tolist() + [idx.numel()]
            runs = [(int(idx[a]), int(idx[b_ - 1]), b_ - a) for a, b_ in zip(cuts[:-1], cuts[1:])]
            print(f"   batch {i}: {neq.numel()} workspace bytes differ from a one-stream trunk; runs (first byte, last byte, count): {runs[:12]}", flush=True)
        who = "TRUNK moved (back-end alone on the slot reproduces the two-stream logits)" if torch.equal(again, outs[i]) else (
            "BACK-END moved (alone on the slot it gives the one-stream logits)" if torch.equal(again, want[i]) else "both differ")
        if stop == 0 or neq.numel():
            line.append((i, float((outs[i] - want[i]).abs().max()) if stop == 0 else int(neq.numel()), who if stop == 0 else "workspace bytes that differ"))
    print(f"{dtype} pass {rep}: {line}", flush=True)
