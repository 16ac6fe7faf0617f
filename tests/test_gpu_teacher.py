"""BASELINE config 3 as a gated configuration: XLS-R (24 layers) + AASIST, batch 16, 4-s clips, fp16 operands, against
the CPU oracle (models/xlsr_aasist.py:86-177), 48 utterances = three batches of 16, the seeded "lively" AASIST head of
tests/golden/make_golden.py (matrices x 1.5).

What is asserted, for EVERY utterance:
  * trunk: SSL features within 1e-3 relative L2 of the fp32 oracle (fp16 operand rounding; measured 7.8e-4);
  * back-end: on the engine's own features the engine equals the oracle back-end to 1e-5 (measured <= 1.2e-6) -- so any
    end-to-end deviation is the REFERENCE MODEL's response to the trunk rounding, nothing of the back-end's;
  * end to end: wherever the reference keeps its GraphPool decisions under that rounding (same node sequences, see
    conftest.teacher_conditioning), |dlogit| <= 1e-3, the `north_star` tolerance.
The reference's top-k is discontinuous (106 score gaps per 4-s utterance, the smallest typically ~5e-6, against a
trunk-induced score perturbation of ~1e-5): on most random-weight utterances SOME pair of near-tied nodes swaps, which
re-pairs nodes in the branch merge and moves a logit by up to ~1e-2 with this head (default-init heads: ~2e-5, they
are insensitive).  That is a property of the model, not a tolerance of this build: the flip statistics are reported
(profiles/r02_teacher_flip_rate.json, tools/teacher_flip_rate.py), exact mode (dtype "fp32") removes the
perturbation altogether (tests/test_gpu_exact.py: 1e-5 on every utterance)."""
import pytest
import torch

from conftest import teacher_conditioning, teacher_oracle

pytestmark = pytest.mark.gpu

N_BATCHES = 3  # x 16 utterances


@pytest.fixture(scope="module")
def oracle48():
    """The 48 utterances of config 3's gate and ONE pass of the CPU oracle over them (24 fp32 layers on the host: the
    expensive half), shared by the fp16 gate and the unconditional-parity modes below."""
    from afx import synth
    sd = synth.model_state_dict("XLSR_AASIST", n_layers=24, head_scale=1.5)
    waves = [torch.cat([synth.waveforms(1, 64000, batch_idx=5000 + 16 * b + i) for i in range(16)]) for b in range(N_BATCHES)]
    return sd, waves, [teacher_oracle(sd, w) for w in waves]


def _rows(oracle48, dtype):
    from afx import engine
    sd, waves, oracles = oracle48
    eng = engine.Engine("xlsr_aasist", n_layers=24, dtype=dtype)
    eng.load_state_dict(sd)
    rows = []
    for wave, orc in zip(waves, oracles):
        _ref, got, r = teacher_conditioning(sd, wave, eng, oracle=orc)
        assert torch.equal(eng.forward(wave.cuda()).cpu(), got)  # launch after launch the same scores
        rows += r
    return rows


def test_config3_teacher_fp16_batch16_every_utterance(oracle48):
    rows = _rows(oracle48, "fp16")
    kept = [r for r in rows if r["same_topk"]]
    there = ["%.1e" % r["dlogit"] for r in kept]
    elsewhere = max((r["dlogit"] for r in rows if not r["same_topk"]), default=0.0)
    print(f"config 3, fp16: {len(kept)} of {len(rows)} utterances keep every top-k decision; |dlogit| there {there}; "
          f"elsewhere max {elsewhere:.1e}")
    assert max(r["feat_rel_l2"] for r in rows) <= 1e-3
    assert max(r["backend"] for r in rows) <= 1e-5
    # profiles/r02_teacher_flip_rate.json (tools/teacher_flip_rate.py, the same 48 utterances and head): 6 of them keep every
    # decision at the build's feature error; the flips are driven by the size of the feature perturbation, so a build whose
    # trunk error grew keeps fewer.  The floor is the recorded count minus one (one near-tie may land either way on another
    # box's summation order), read from the file rather than written here.
    import json
    import os
    rec = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02_teacher_flip_rate.json")))
    floor = rec["lively_1.5"]["keep_every_topk_decision"] - 1
    assert len(kept) >= floor, f"{len(kept)} utterances keep their decisions, the recorded build kept {floor + 1}"
    assert all(r["dlogit"] <= 1e-3 for r in kept), [r for r in kept if r["dlogit"] > 1e-3]
    # where a near-tied pair of nodes does swap, the logit moves by what the reference model makes of the swap: measured
    # <= 9.1e-3 with this head; a coarse absolute bound still catches a back-end that breaks only when the order changes
    assert elsewhere <= 3e-2, elsewhere


@pytest.mark.parametrize("dtype", ["fp32", "fp16x3"])
def test_config3_teacher_unconditional_parity_modes(oracle48, dtype):
    """The answer to "scores within 1e-3 on EVERY utterance" for config 3: the SAME 48 utterances,
    24 layers and lively head as the fp16 gate above, in the modes that carry fp32 accuracy through the trunk -- every
    utterance within the `north_star` tolerance and every GraphPool decision identical to the oracle's, no conditioning
    on which utterances happen not to flip."""
    rows = _rows(oracle48, dtype)
    worst = max(r["dlogit"] for r in rows)
    flips = [i for i, r in enumerate(rows) if not r["same_topk"]]
    print(f"config 3, {dtype}: 48 utterances, max |dlogit| {worst:.2e}, feature rel L2 max {max(r['feat_rel_l2'] for r in rows):.1e}, "
          f"utterances with a changed top-k decision: {flips}")
    assert len(rows) == 16 * N_BATCHES
    assert worst <= 1e-3, [(i, r["dlogit"]) for i, r in enumerate(rows) if r["dlogit"] > 1e-3]
    assert not flips, flips
    assert max(r["backend"] for r in rows) <= 1e-5
