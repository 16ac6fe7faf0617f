"""BASELINE config 3 (XLS-R-24 + AASIST, fp16, 4-s clips): how often does the reference model's own top-k flip under
the fp16 trunk's rounding, and what does a flip cost -- the numbers DESIGN.md quotes and tests/test_gpu_teacher.py
explains (run on the GPU box; writes a JSON, copy it to profiles/r02_teacher_flip_rate.json).

Per utterance: same_topk (the oracle back-end picks identical GraphPool node sequences on its own fp32 SSL features
and on the GPU's fp16-trunk features), the smallest deciding score gap, |dlogit| end to end, |dlogit| of the back-end
alone (GPU vs oracle back-end on the GPU's features) and the relative L2 error of the features.  Two heads: the
seeded "lively" head (matrices x 1.5, tests/golden/make_golden.py) and the default-init one."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd"), os.path.join(ROOT, "tests")]
from afx import engine, synth  # noqa: E402
from conftest import teacher_conditioning  # noqa: E402

SEED0 = 5000


def survey(head_scale, seeds, n_layers=24, dtype="fp16"):
    sd = synth.model_state_dict("XLSR_AASIST", n_layers=n_layers, head_scale=head_scale)
    eng = engine.Engine("xlsr_aasist", n_layers=n_layers, dtype=dtype)
    eng.load_state_dict(sd)
    rows = []
    for i in range(0, len(seeds), 16):
        chunk = seeds[i:i + 16]
        wave = torch.cat([synth.waveforms(1, 64000, batch_idx=s) for s in chunk])
        _ref, _got, r = teacher_conditioning(sd, wave, eng)
        for s, row in zip(chunk, r):
            row["seed"] = s
        rows += r
        print(f"head_scale {head_scale}: {i + len(chunk)}/{len(seeds)} utterances", flush=True)
    return rows


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/teacher_flip_rate.json"
    n_cand = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    seeds = list(range(SEED0, SEED0 + n_cand))
    t0 = time.time()
    res = {}
    for name, scale in (("lively_1.5", 1.5), ("default_init", None)):
        rows = survey(scale, seeds)
        kept = [r for r in rows if r["same_topk"]]
        flipped = [r for r in rows if not r["same_topk"]]
        res[name] = dict(
            utterances=len(rows), keep_every_topk_decision=len(kept), topk_flips=len(flipped),
            over_1e3=sum(r["dlogit"] > 1e-3 for r in rows), over_1e3_among_kept=sum(r["dlogit"] > 1e-3 for r in kept),
            max_dlogit_kept=max((r["dlogit"] for r in kept), default=None),
            max_dlogit_flipped=max((r["dlogit"] for r in flipped), default=None),
            max_backend_alone=max(r["backend"] for r in rows),
            median_feat_rel_l2=sorted(r["feat_rel_l2"] for r in rows)[len(rows) // 2],
            median_smallest_gap=sorted(r["margin"] for r in rows)[len(rows) // 2], rows=rows)
        print(name, {k: v for k, v in res[name].items() if k != "rows"}, flush=True)
    res["seconds"] = time.time() - t0
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
