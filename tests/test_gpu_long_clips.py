"""Clips longer than the 4-s headline configuration.  `test_duration_sec` is a free value of the reference's
config (config.py:75, data/test_set.py:16,78,153,255,347), so the drop-in has to score 8-s, 10-s, 30-s clips
too: the trunk attention, the Shaw attention and the graph kernels switch to their blocked / dynamically sized
forms (DESIGN.md "Clip length"), everything else is length-agnostic.  Same contract as test_gpu_models.py:
oracle on the same seeded weights and waveforms, |dlogit| <= 1e-3 with fp16 operands."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SCORE_TOL = 1e-3


def rel_l2(a, b):
    a, b = a.float().cpu().reshape(-1), b.float().cpu().reshape(-1)
    return ((a - b).norm() / b.norm()).item()


@pytest.fixture(scope="module")
def afx_mod():
    import afx  # noqa: F401
    from afx import engine, synth
    return engine, synth


@pytest.mark.parametrize("L,T,B", [(160000, 499, 2), (72000, 224, 2), (72320, 225, 2), (480000, 1499, 1)])
def test_ssl_trunk_on_long_clips(afx_mod, L, T, B):
    """10-s and 30-s clips, and the two lengths either side of the one-pass / blocked attention switch."""
    engine, synth = afx_mod
    from oracle import ssl_trunk
    sd = synth.ssl_state_dict(1)
    wave = synth.waveforms(B, L, batch_idx=L % 997)
    ref = ssl_trunk.ssl_forward({k[len(synth.SSL_PREFIX):]: v for k, v in sd.items()}, wave)
    eng = engine.Engine("ssl", n_layers=1, dtype="fp16")
    eng.load_state_dict(sd)
    got = eng.ssl(wave.cuda())
    assert got.shape == (B, T, 1024) == ref.shape
    assert rel_l2(got, ref) < 2e-3


def test_conformer_student_on_8s_clips(afx_mod):
    """N = 400 tokens: Shaw attention leaves the 209-token matrix-core kernel for the blocked fp32 kernel."""
    engine, synth = afx_mod
    from oracle import models
    sd = synth.model_state_dict("ConformerModel", n_layers=2)
    wave = synth.waveforms(3, 128000, batch_idx=8)
    taps = {}
    ref = models.conformer_forward(sd, wave, taps=taps)
    eng = engine.Engine("conformer", n_layers=2, dtype="fp16")
    eng.load_state_dict(sd)
    eng.enable_taps()
    got = eng.forward(wave.cuda()).cpu()
    assert taps["ssl"].shape[1] == 399
    assert rel_l2(eng.tap("ssl"), taps["ssl"]) < 2e-3
    for b in range(4):
        assert rel_l2(eng.tap(f"block{b}"), taps[f"block{b}"]) < 3e-3
    err = (got - ref).abs().max().item()
    assert err <= SCORE_TOL, f"max |dlogit| {err:.3e}"
    exact = engine.Engine("conformer", n_layers=2, dtype="fp32")
    exact.load_state_dict(sd)
    e32 = (exact.forward(wave.cuda()).cpu() - ref).abs().max().item()
    print(f"8-s student: fp16 max|dlogit| {err:.2e}, fp32 exact mode {e32:.2e}")
    assert e32 <= 2e-5


def test_conformer_student_on_a_30s_clip(afx_mod):
    """T = 1499: every length-dependent kernel in its blocked form at once (1 trunk layer, 1 block)."""
    engine, synth = afx_mod
    from oracle import models
    sd = synth.model_state_dict("ConformerModel", n_layers=1, n_encoders=1)
    wave = synth.waveforms(1, 480000, batch_idx=30)
    ref = models.conformer_forward(sd, wave)
    eng = engine.Engine("conformer", n_layers=1, dtype="fp16", conf_blocks=1)
    eng.load_state_dict(sd)
    err = (eng.forward(wave.cuda()).cpu() - ref).abs().max().item()
    assert err <= SCORE_TOL, f"max |dlogit| {err:.3e}"


def test_conformer_head_beyond_the_relative_distance_clamp(afx_mod):
    """T = 700 frames (14 s): distances beyond max_pos_emb = 512 share the edge embeddings (lucidrains
    conformer Attention: dist.clamp(-max_pos_emb, max_pos_emb))."""
    engine, synth = afx_mod
    from oracle import conformer, models
    sd = synth.model_state_dict("ConformerModel", n_layers=1)
    feats = torch.randn(2, 700, 1024, generator=torch.Generator().manual_seed(14))
    ref = conformer.conformer_head(models.split(sd)[1], feats, heads=4)
    eng = engine.Engine("conformer", n_layers=1, dtype="fp16")
    eng.load_state_dict(sd)
    got = eng.head(feats.cuda()).cpu()
    err = (got - ref).abs().max().item()
    assert err <= SCORE_TOL, f"max |dlogit| {err:.3e}"


def test_teacher_on_8s_clips(afx_mod):
    """AASIST graphs grow with the clip (T // 3 = 133 temporal nodes): the graph kernels size their LDS per
    launch.  Contract as in test_gpu_aasist.py: the fp32 back-end reproduces the oracle on the oracle's own
    features to 1e-5 (every top-k decision agrees); exact mode end to end within 1e-4."""
    engine, synth = afx_mod
    from oracle import models
    sd = synth.model_state_dict("XLSR_AASIST", n_layers=1)
    wave = synth.waveforms(3, 128000, batch_idx=5)
    taps = {}
    ref = models.xlsr_aasist_forward(sd, wave, taps=taps)
    eng = engine.Engine("xlsr_aasist", n_layers=1, dtype="fp16")
    eng.load_state_dict(sd)
    exact_head = eng.head(taps["ssl"].cuda()).cpu()
    assert (exact_head - ref).abs().max().item() <= 1e-5
    got = eng.forward(wave.cuda()).cpu()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max(dim=1)[0].median().item() <= 1e-3
    e32 = engine.Engine("xlsr_aasist", n_layers=1, dtype="fp32")
    e32.load_state_dict(sd)
    assert (e32.forward(wave.cuda()).cpu() - ref).abs().max().item() <= 1e-4


def test_clip_too_long_for_the_graph_kernels_is_a_loud_error(afx_mod):
    engine, synth = afx_mod
    from afx._lib import AfxError
    sd = synth.model_state_dict("XLSR_AASIST", n_layers=1)
    eng = engine.Engine("xlsr_aasist", n_layers=1, dtype="fp16")
    eng.load_state_dict(sd)
    with pytest.raises(AfxError, match="too long"):
        eng.head(torch.zeros(1, 1900, 1024, device="cuda"))


@pytest.mark.parametrize("B,L", [(1, 63999), (5, 401), (65, 16000), (3, 100003), (2, 24321)])
def test_student_on_odd_batch_and_clip_sizes(afx_mod, B, L):
    """Batch sizes and sample counts that are multiples of nothing (tile tails everywhere: 1 frame at L = 401,
    B·T not a multiple of any tile, the blocked attention edge at T = 312)."""
    engine, synth = afx_mod
    from oracle import models
    sd = synth.model_state_dict("ConformerModel", n_layers=1, n_encoders=1)
    wave = synth.waveforms(B, L, batch_idx=L % 1000)
    ref = models.conformer_forward(sd, wave)
    eng = engine.Engine("conformer", n_layers=1, dtype="fp16", conf_blocks=1)
    eng.load_state_dict(sd)
    got = eng.forward(wave.cuda()).cpu()
    err = (got - ref).abs().max().item()
    assert err <= SCORE_TOL, f"B={B} L={L}: max |dlogit| {err:.3e}"
    again = eng.forward(wave.cuda()).cpu()
    assert torch.equal(got, again)


def test_bf16_operands_on_a_long_clip(afx_mod):
    engine, synth = afx_mod
    from oracle import models
    sd = synth.model_state_dict("ConformerModel", n_layers=1, n_encoders=1)
    wave = synth.waveforms(2, 128000, batch_idx=77)
    ref = models.conformer_forward(sd, wave)
    eng = engine.Engine("conformer", n_layers=1, dtype="bf16", conf_blocks=1)
    eng.load_state_dict(sd)
    err = (eng.forward(wave.cuda()).cpu() - ref).abs().max().item()
    assert err < 3e-2, f"bf16 max |dlogit| {err:.3e}"  # bf16 is measured, not gated at 1e-3 (DESIGN.md numerics)
