"""BASELINE config 3 parity fixture (XLS-R-24 + AASIST, fp16, 4-s clips): choose 16 utterances on which the reference
model itself is well-conditioned, and count how often it is not.

The AASIST back-end is discontinuous: GraphPool keeps the top half of the nodes in descending score order
(models/aasist_modules.py:330-336) and the two branches are merged position by position (models/xlsr_aasist.py:160-162),
so a perturbation of the node scores larger than the smallest gap between adjacent kept scores re-pairs nodes and moves a
logit by ~1e-2 -- in the reference's own fp32 arithmetic as much as in ours.  An utterance is WELL-CONDITIONED for the
1e-3 contract when, for all six pools,
    (a) the oracle's top-k index sequences on its own fp32 SSL features and on the GPU's fp16-trunk SSL features agree, and
    (b) the smallest deciding gap is >= 3e-5 on BOTH feature sets (i.e. still there AFTER the fp16 trunk error).
Run on the GPU box:  python tools/pick_teacher_fixture.py gpurun_out/teacher_fixture.json [candidates]
The 16 chosen waveform seeds go to tests/golden/teacher_b16.json (test_gpu_teacher.py recomputes every number in it);
the flip statistics (lively head and default-init head) are the "flip rate" DESIGN.md quotes.
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import engine, synth  # noqa: E402
from oracle import aasist as oa  # noqa: E402
from oracle import models as om  # noqa: E402

MARGIN = 3e-5
SEED0 = 5000


def wave_of(seed):
    return synth.waveforms(1, 64000, batch_idx=seed)


def survey(head_scale, seeds, n_layers=24, dtype="fp16"):
    sd = synth.model_state_dict("XLSR_AASIST", n_layers=n_layers, head_scale=head_scale)
    _ssl, head = om.split(sd)
    eng = engine.Engine("xlsr_aasist", n_layers=n_layers, dtype=dtype)
    eng.load_state_dict(sd)
    eng.enable_taps()
    rows = []
    for i in range(0, len(seeds), 16):
        chunk = seeds[i:i + 16]
        wave = torch.cat([wave_of(s) for s in chunk])
        t_ref = {}
        ref = om.xlsr_aasist_forward(sd, wave, taps=t_ref)
        got = eng.forward(wave.cuda()).cpu()
        feats = eng.tap("ssl").cpu().reshape(t_ref["ssl"].shape)
        t_mid = {}
        mid = oa.aasist_backend(head, feats, t_mid)
        for j, s in enumerate(chunk):
            same = all(torch.equal(t_ref["pool_idx"][p][j], t_mid["pool_idx"][p][j]) for p in t_ref["pool_idx"])
            m_ref = min(float(t_ref["pool_margin"][p][j]) for p in t_ref["pool_margin"])
            m_mid = min(float(t_mid["pool_margin"][p][j]) for p in t_mid["pool_margin"])
            rows.append(dict(seed=s, same_topk=bool(same), margin_ref=m_ref, margin_gpu_feats=m_mid,
                             dlogit=float((got[j] - ref[j]).abs().max()),
                             dlogit_backend_only=float((got[j] - mid[j]).abs().max()),
                             feat_rel_l2=float((feats[j] - t_ref["ssl"][j]).norm() / t_ref["ssl"][j].norm())))
        print(f"head_scale {head_scale}: {i + len(chunk)}/{len(seeds)} utterances", flush=True)
    return rows


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/teacher_fixture.json"
    n_cand = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    seeds = list(range(SEED0, SEED0 + n_cand))
    t0 = time.time()
    res = {}
    for name, scale in (("lively_1.5", 1.5), ("default_init", None)):
        rows = survey(scale, seeds)
        ok = [r for r in rows if r["same_topk"] and min(r["margin_ref"], r["margin_gpu_feats"]) >= MARGIN]
        flipped = [r for r in rows if not r["same_topk"]]
        res[name] = dict(
            utterances=len(rows), well_conditioned=len(ok), topk_flips=len(flipped),
            over_1e3=sum(r["dlogit"] > 1e-3 for r in rows),
            over_1e3_among_well_conditioned=sum(r["dlogit"] > 1e-3 for r in ok),
            max_dlogit_well_conditioned=max((r["dlogit"] for r in ok), default=None),
            max_dlogit_flipped=max((r["dlogit"] for r in flipped), default=None),
            median_feat_rel_l2=sorted(r["feat_rel_l2"] for r in rows)[len(rows) // 2], rows=rows)
        print(name, {k: v for k, v in res[name].items() if k != "rows"}, flush=True)
    ok = [r for r in res["lively_1.5"]["rows"] if r["same_topk"] and min(r["margin_ref"], r["margin_gpu_feats"]) >= MARGIN]
    res["fixture"] = dict(head_scale=1.5, n_layers=24, dtype="fp16", margin=MARGIN, seeds=[r["seed"] for r in ok[:16]],
                          margin_ref=[r["margin_ref"] for r in ok[:16]],
                          margin_gpu_feats=[r["margin_gpu_feats"] for r in ok[:16]],
                          dlogit_when_picked=[r["dlogit"] for r in ok[:16]])
    res["seconds"] = time.time() - t0
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    json.dump(res, open(out, "w"), indent=1)
    print("fixture seeds:", res["fixture"]["seeds"], "->", out)


if __name__ == "__main__":
    main()
