"""Checkpoints with XLS-R-style OUTLIER CHANNELS ("Missing 2"): every other parity number of this suite is
on seeded weights at init scale, where all activations are O(1); trained wav2vec2 / XLS-R trunks carry a few channels orders of
magnitude above the rest.  ``afx.synth.with_outliers`` scales, in every transformer layer, a few rows of fc1 / out_proj and a
few LayerNorm gains by ``gain``: with 30 - 1000 the residual stream reaches 5e1 - 1e5 and the FFN hidden follows (the
reference loads such a trunk at models/fe.py:11-21).  What each precision does with it, against the fp32 CPU oracle:

  fp16x3  every score within 1e-5 up to gain 1000 (residual 1e5) -- the same as exact mode; its LayerNorm outputs take a
          per-LayerNorm power-of-two scale chosen at finalize from the gains, so they cannot leave fp16 whatever the
          checkpoint, and every other operand copy has fp16's own range (65 504);
  fp16    the Conformer student holds 1e-3 up to gain 1000; the teacher's lively GraphPool head holds it to gain 30 and moves by
          what the reference makes of swapped near-ties beyond (<= 3e-2: the statement of tests/test_gpu_teacher.py);
  both    an operand copy that DOES leave the format (gain 3e4: FFN hidden beyond 65 504) is an ``AfxError`` from
          ``check_finite()`` -- which the scoring loops call before they write a score -- never a NaN or a plausible-looking
          score in a file.  Exact mode (fp32) stays finite and accurate there.
Measured table: profiles/r04_outliers_diag.txt (tools/diag_outliers.py)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

NL, B, L = 4, 4, 16000
MODELS = {"conformer": ("ConformerModel", dict(n_encoders=2), dict(conf_blocks=2)),
          "xlsr_aasist": ("XLSR_AASIST", dict(head_scale=1.5), {})}


def _case(arch, gain):
    from afx import synth
    from oracle import models as om
    name, kw, _ = MODELS[arch]
    sd = synth.with_outliers(synth.model_state_dict(name, n_layers=NL, **kw), gain=gain)
    wave = synth.waveforms(B, L, batch_idx=77)
    taps = {}
    ref = (om.conformer_forward if arch == "conformer" else om.xlsr_aasist_forward)(sd, wave, taps=taps)
    resid = max(float(taps[f"layer{n}"].abs().max()) for n in range(NL))
    return sd, wave, ref, resid


def _engine(arch, dtype, sd):
    from afx import engine
    eng = engine.Engine(arch, n_layers=NL, dtype=dtype, **MODELS[arch][2])
    eng.load_state_dict(sd)
    return eng


@pytest.mark.parametrize("arch", ["conformer", "xlsr_aasist"])
@pytest.mark.parametrize("gain", [30.0, 300.0, 1000.0])
def test_scores_on_outlier_checkpoints(arch, gain):
    sd, wave, ref, resid = _case(arch, gain)
    assert resid > (40 if gain == 30 else 5e3)  # the fixture does what it says: residual values of 1e2 .. 1e5
    err = {}
    for dtype in ("fp16", "fp16x3", "fp32"):
        eng = _engine(arch, dtype, sd)
        got = eng.forward(wave.cuda()).cpu()
        eng.check_finite()  # nothing left its format at these gains, in any precision
        err[dtype] = (got - ref).abs().max().item()
    print(f"{arch} gain {gain:g}: oracle max |residual| {resid:.3g}; max |dlogit| fp16 {err['fp16']:.2e}, fp16x3 {err['fp16x3']:.2e}, fp32 {err['fp32']:.2e}")
    # split precision = exact mode's accuracy (at 1e5 the fp32 ORACLE's own rounding is 1e-5 of a logit: both sit on it)
    assert err["fp16x3"] <= (1e-5 if gain < 1000 else 1e-4) and err["fp32"] <= (1e-5 if gain < 1000 else 1e-4)
    if arch == "conformer" or gain <= 30:
        assert err["fp16"] <= 1e-3
    else:  # lively GraphPool head: near-ties swap under the fp16 trunk's larger error; what the reference makes of a swap
        assert err["fp16"] <= 3e-2


@pytest.mark.parametrize("dtype", ["fp16", "fp16x3"])
def test_an_operand_overflow_is_a_loud_error_not_a_score(dtype, tmp_path):
    from afx import harness, synth
    from afx._lib import AfxError
    from models.conformer_baseline import MyModel
    sd, wave, ref, resid = _case("conformer", 3e4)
    assert torch.isfinite(ref).all() and resid > 1e7
    eng = _engine("conformer", dtype, sd)
    eng.forward(wave.cuda())
    with pytest.raises(AfxError, match="non-finite"):
        eng.check_finite()
    eng.check_finite()  # the counters were cleared by the failed check
    # the same engine on a healthy batch of a healthy checkpoint is not poisoned
    ok_sd = synth.model_state_dict("ConformerModel", n_layers=NL, n_encoders=2)
    eng.load_state_dict(ok_sd)
    eng.forward(wave.cuda())
    eng.check_finite()
    # exact mode computes the overflowing checkpoint like the oracle does
    ex = _engine("conformer", "fp32", sd)
    got = ex.forward(wave.cuda()).cpu()
    ex.check_finite()
    assert (got - ref).abs().max().item() <= 1e-3 * max(1.0, float(ref.abs().max()))
    # the scoring loop refuses to write a score file from overflowed batches
    m = MyModel(device="cuda", ssl_cpkt_path=None, num_layers=NL, order="first", n_encoders=2).to("cuda").eval()
    m.load_state_dict(sd)
    m.set_precision(dtype)

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 4

        def __getitem__(self, i):
            return f"utt{i}", wave[i], 1

    path = str(tmp_path / "scores.txt")
    with pytest.raises(AfxError, match="non-finite"):
        harness.produce_evaluation_file(DS(), m, "cuda", path, batch_size=2, num_workers=0)
    assert not os.path.exists(path)
