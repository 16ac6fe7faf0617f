"""End-to-end parity on a real MI355X: the native engine (through the C ABI)
against the CPU oracle on the same seeded weights and waveforms.

Tolerances (written here as the contract):
  * logits / bonafide score:  |delta| <= 1e-3   (north_star), fp16 operands
  * intermediate features:    relative L2 error <= 2e-3 (fp16), 2e-2 (bf16)
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

SCORE_TOL = 1e-3


def rel_l2(a, b):
    a, b = a.float().cpu().reshape(-1), b.float().cpu().reshape(-1)
    return ((a - b).norm() / b.norm()).item()


@pytest.fixture(scope="module")
def afx_mod():
    import afx
    from afx import engine, synth
    return engine, synth


def _ssl_sd(synth, n_layers):
    return synth.ssl_state_dict(n_layers)


@pytest.mark.parametrize("dtype,tol", [("fp16", 2e-3), ("bf16", 2e-2)])
def test_ssl_trunk_stage_by_stage(afx_mod, dtype, tol):
    engine, synth = afx_mod
    from oracle import ssl_trunk
    sd = _ssl_sd(synth, 2)
    wave = synth.waveforms(2, 64000)
    taps = {}
    ref = ssl_trunk.ssl_forward({k[len(synth.SSL_PREFIX):]: v for k, v in sd.items()}, wave, taps=taps)
    eng = engine.Engine("ssl", n_layers=2, dtype=dtype)
    eng.load_state_dict(sd)
    eng.enable_taps()
    got = eng.ssl(wave.cuda())
    assert got.shape == (2, 199, 1024)
    for name in ("conv", "proj", "pos", "layer0", "layer1"):
        e = rel_l2(eng.tap(name), taps[name])
        assert e < tol, f"{name}: rel L2 {e:.3e}"
    assert rel_l2(got, ref) < tol
    if dtype == "fp16":
        assert (got.cpu() - ref).abs().max().item() < 0.05


@pytest.mark.parametrize("L,T", [(16000, 49), (64600, 201), (4000, 12), (400, 1)])
def test_ssl_trunk_other_clip_lengths(afx_mod, L, T):
    engine, synth = afx_mod
    from oracle import ssl_trunk
    sd = _ssl_sd(synth, 1)
    wave = synth.waveforms(3, L, batch_idx=L)
    ref = ssl_trunk.ssl_forward({k[len(synth.SSL_PREFIX):]: v for k, v in sd.items()}, wave)
    eng = engine.Engine("ssl", n_layers=1, dtype="fp16")
    eng.load_state_dict(sd)
    got = eng.ssl(wave.cuda())
    assert got.shape == (3, T, 1024) == ref.shape
    assert rel_l2(got, ref) < 2e-3
    got3 = eng.ssl(wave.cuda().unsqueeze(-1))  # (B,L,1) like models/fe.py:18
    assert torch.equal(got3, got)


def test_pre_emphasis_is_fused_into_the_first_kernel(afx_mod):
    engine, synth = afx_mod
    from oracle import pre, ssl_trunk
    sd = _ssl_sd(synth, 1)
    wave = synth.waveforms(2, 16000, batch_idx=3)
    ref = ssl_trunk.ssl_forward({k[len(synth.SSL_PREFIX):]: v for k, v in sd.items()}, pre.pre_emphasis(wave))
    eng = engine.Engine("ssl", n_layers=1, dtype="fp16", pre_emphasis=True)
    eng.load_state_dict(sd)
    assert rel_l2(eng.ssl(wave.cuda()), ref) < 2e-3


def test_conformer_student_scores_match_oracle(afx_mod):
    """BASELINE config 2 shape: XLS-R first-6 trunk + 4 Conformer blocks (emb 144)."""
    engine, synth = afx_mod
    from oracle import models
    sd = synth.model_state_dict("ConformerModel", n_layers=6)
    wave = synth.waveforms(6, 64000)
    taps = {}
    ref = models.conformer_forward(sd, wave, taps=taps)
    eng = engine.Engine("conformer", n_layers=6, dtype="fp16")
    eng.load_state_dict(sd)
    eng.enable_taps()
    got = eng.forward(wave.cuda()).cpu()
    assert rel_l2(eng.tap("ssl"), taps["ssl"]) < 2e-3
    assert rel_l2(eng.tap("tokens"), taps["tokens"]) < 2e-3
    for b in range(4):
        assert rel_l2(eng.tap(f"block{b}"), taps[f"block{b}"]) < 3e-3
    err = (got - ref).abs().max().item()
    assert err <= SCORE_TOL, f"max |dlogit| {err:.3e}: {got} vs {ref}"
    # bf16 operands are measured, not gated at 1e-3 (DESIGN.md numerics)
    eb = engine.Engine("conformer", n_layers=6, dtype="bf16")
    eb.load_state_dict(sd)
    errb = (eb.forward(wave.cuda()).cpu() - ref).abs().max().item()
    print(f"conformer student: fp16 max|dlogit| {err:.2e}, bf16 {errb:.2e}")
    assert errb < 3e-2


def test_conformer_head_alone_and_small_kernel_size(afx_mod):
    engine, synth = afx_mod
    from oracle import conformer
    head = synth.conformer_head_state_dict(emb_size=144, heads=4, kernel_size=16, n_encoders=2)
    sd = dict(synth.ssl_state_dict(1))
    sd.update(head)
    g = torch.Generator().manual_seed(9)
    feats = torch.randn(3, 49, 1024, generator=g)
    ref = conformer.conformer_head(head, feats, heads=4)
    eng = engine.Engine("conformer", n_layers=1, dtype="fp16", conf_kernel=16, conf_blocks=2)
    eng.load_state_dict(sd)
    got = eng.head(feats.cuda()).cpu()
    assert (got - ref).abs().max().item() <= SCORE_TOL


def test_myconformer_forward_alone(afx_mod):
    """models/conformer_baseline.py:22-29: ``MyConformer.forward(x)`` as a module of its own -- class token, the cloned
    Conformer blocks, ``(fc5(token 0), token 0)`` -- through afx_conformer_forward, against the oracle's restatement of
    the same lines; and the same call on the Conformer blocks of a full student (``model.conformer(tokens)``)."""
    engine, synth = afx_mod
    from models.conformer_baseline import MyConformer, MyModel
    from oracle import conformer as oconf
    head = synth.conformer_head_state_dict(emb_size=144, heads=4, kernel_size=31, n_encoders=2)
    own = {k[len("conformer."):]: v for k, v in head.items() if k.startswith("conformer.")}
    m = MyConformer(emb_size=144, heads=4, kernel_size=31, n_encoders=2).to("cuda").eval()
    m.load_state_dict(own)
    g = torch.Generator().manual_seed(21)
    x = torch.randn(3, 49, 144, generator=g)
    ref_out, ref_emb = oconf.my_conformer(own, x, heads=4, prefix="")
    with torch.no_grad():
        out, emb = m(x.cuda())
    assert out.shape == (3, 2) and emb.shape == (3, 144)
    assert (out.cpu() - ref_out).abs().max().item() <= SCORE_TOL
    assert rel_l2(emb, ref_emb) < 3e-3
    with torch.no_grad():
        out2, _ = m(x.cuda(), "cuda")  # the stray positional argument of MyModel.forward's call (:98)
    assert torch.equal(out2, out)
    with pytest.raises(ValueError, match="tokens"):
        m(torch.zeros(2, 10, 100, device="cuda"))
    with pytest.raises(RuntimeError, match="eval"):
        m.train()(x.cuda())
    # the blocks inside a whole model, called on their own like the reference's `self.conformer(x)` (:63)
    full = MyModel(device="cuda", ssl_cpkt_path=None, num_layers=1, order="first", n_encoders=2).to("cuda").eval()
    full.conformer.load_state_dict(own)
    with torch.no_grad():
        o3, e3 = full.conformer(x.cuda())
    assert torch.equal(o3, out) and torch.equal(e3, emb)


def test_errors_are_loud(afx_mod):
    engine, synth = afx_mod
    from afx._lib import AfxError
    with pytest.raises(AfxError, match="at least 1 and at most 24"):
        engine.Engine("ssl", n_layers=0)
    eng = engine.Engine("ssl", n_layers=1)
    with pytest.raises(AfxError, match="not finalized"):
        eng.ssl(torch.zeros(1, 16000, device="cuda"))
    sd = _ssl_sd(synth, 1)
    bad = {k: v for k, v in sd.items() if "fc2.bias" not in k}
    with pytest.raises(AfxError, match="missing weight"):
        eng.load_state_dict(bad)
    eng.load_state_dict(sd)
    with pytest.raises(AfxError, match="too few"):
        eng.ssl(torch.zeros(1, 300, device="cuda"))
    with pytest.raises(AfxError, match="GPU"):
        eng.ssl(torch.zeros(1, 16000))


def test_hipgraph_replay_matches_eager_and_harness_writes_scores(afx_mod, tmp_path):
    """The forward allocates nothing and never synchronises, so it captures into a hipGraph;
    replay must be bit-identical to the eager launch sequence.  Also the scoring loop
    (main.py:199-221 semantics) end to end on a toy dataset."""
    engine, synth = afx_mod
    from afx import harness
    sd = synth.model_state_dict("ConformerModel", n_layers=1, n_encoders=1)
    eng = engine.Engine("conformer", n_layers=1, dtype="fp16", conf_blocks=1)
    eng.load_state_dict(sd)
    wave = synth.waveforms(2, 16000, batch_idx=11).cuda()
    eager = eng.forward(wave).clone()
    run = eng.capture(2, 16000)
    assert torch.equal(run(wave), eager)
    wave2 = synth.waveforms(2, 16000, batch_idx=12).cuda()
    assert torch.equal(run(wave2), eng.forward(wave2))

    class Toy(torch.utils.data.Dataset):
        def __len__(self):
            return 5

        def __getitem__(self, i):
            return f"utt{i}", synth.waveforms(1, 16000, batch_idx=100 + i)[0], 1

    class Wrap(torch.nn.Module):
        def forward(self, x):
            return eng.forward(x)

    path = tmp_path / "scores" / "toy.txt"
    names, scores = harness.produce_evaluation_file(Toy(), Wrap(), "cuda", str(path), batch_size=2, num_workers=0)
    lines = path.read_text().strip().split("\n")
    assert names == [f"utt{i}" for i in range(5)] and len(lines) == 5
    ref = torch.cat([eng.forward(synth.waveforms(1, 16000, batch_idx=100 + i).cuda())[:, 1] for i in range(5)]).cpu()
    got = torch.tensor([float(l.split()[1]) for l in lines])
    assert lines[0].split()[0] == "utt0" and (got - ref).abs().max().item() < 1e-6
    pe = harness.PreEmphasis(coef=0.97, enabled=True)
    from oracle import pre
    x = synth.waveforms(3, 1000, batch_idx=5)
    assert (pe(x.cuda()).cpu() - pre.pre_emphasis(x)).abs().max().item() < 1e-6


def test_eer_of_the_build_matches_the_oracle(afx_mod):
    """EER delta on a synthetic trial list.  Labels are drawn from the oracle's
    own scores plus noise (so the oracle EER sits at 10-20 %); the fp16 engine's scores must rank
    the trials the same way: |EER_build - EER_oracle| < 0.005 percentage points."""
    engine, synth = afx_mod
    from afx import harness
    from oracle import models, pre
    sd = synth.model_state_dict("ConformerModel", n_layers=2, n_encoders=2)
    eng = engine.Engine("conformer", n_layers=2, dtype="fp16", conf_blocks=2)
    eng.load_state_dict(sd)
    ref, got = [], []
    for i in range(8):  # 256 one-second trials
        wave = synth.waveforms(32, 16000, batch_idx=500 + i)
        ref.append(models.conformer_forward(sd, wave)[:, 1])
        got.append(eng.forward(wave.cuda())[:, 1].cpu())
    ref, got = torch.cat(ref), torch.cat(got)
    g = torch.Generator().manual_seed(4096)
    noisy = ref + ref.std() * 0.8 * torch.randn(ref.shape, generator=g)
    labels = (noisy > noisy.median()).long().numpy()
    eer_ref = pre.eer_percent(ref.numpy(), labels)
    eer_got = harness.calculate_EER(got.numpy(), labels)
    print(f"EER oracle {eer_ref:.4f} %  build {eer_got:.4f} %  max|dscore| {(got - ref).abs().max().item():.2e}")
    assert 5.0 < eer_ref < 35.0
    assert abs(eer_got - eer_ref) < 0.005


def test_benchmarked_student_4096_trials_scores_and_eer(afx_mod):
    """SURVEY.md 8(d) on the benchmarked configuration itself (BASELINE configs[1]: first-6 XLS-R trunk + 4 Conformer
    blocks, fp16, batch 64, 4-s clips), 4096 trials against the CPU oracle (committed fixture
    tests/golden/eer_student_4096.npz, made by tools/make_eer_fixture.py -- the oracle needs 12 minutes for them):

      * EVERY bonafide score within the 1e-3 tolerance of `north_star` (this is the 4096-utterance parity sweep);
      * the EER: "unchanged to 2 d.p." presupposes scores spread far wider than the tolerance (trained checkpoints:
        logits over several units).  The random-init student's scores have std 0.04 -- the median gap between
        neighbouring scores is a few 1e-5, far inside the tolerance -- so WHICH of two near-tied opposite-label trials
        ranks first at the operating point is below the contract's resolution, and one such swap moves the EER by
        1/2048 = 0.049 percentage points.  Asserted therefore: the build's EER lies inside the band the oracle's own
        scores span under an adversarial +-1e-3 perturbation (1.7 pp wide here), and within four trials (0.2 pp) of the oracle's
        EER (measured: three, 0.146 pp)."""
    engine, synth = afx_mod
    from conftest import load_golden
    from afx import harness
    z = load_golden("eer_student_4096.npz")
    n_batches = z["scores"].shape[0] // 64
    sd = synth.model_state_dict("ConformerModel", n_layers=6)
    eng = engine.Engine("conformer", n_layers=6, dtype="fp16")
    eng.load_state_dict(sd)
    got = torch.cat([eng.forward(synth.waveforms(64, 64000, batch_idx=9000 + i).cuda())[:, 1].cpu() for i in range(n_batches)])
    ref = torch.from_numpy(z["scores"])
    labels = z["labels"].astype(int)
    d = (got - ref).abs()
    eer_ref = float(z["eer"])
    eer_got = harness.calculate_EER(got.numpy(), labels)
    sign = torch.from_numpy(labels * 2 - 1).float()
    eer_lo = harness.calculate_EER((ref + SCORE_TOL * sign).numpy(), labels)  # every trial pushed the right way
    eer_hi = harness.calculate_EER((ref - SCORE_TOL * sign).numpy(), labels)  # ... the wrong way
    print(f"{len(labels)} trials: max|dscore| {d.max().item():.2e} mean {d.mean().item():.2e}; EER oracle {eer_ref:.4f} % build {eer_got:.4f} % "
          f"(band of the +-1e-3 tolerance itself: {eer_lo:.4f} .. {eer_hi:.4f} %; score std {ref.std().item():.3f})")
    assert d.max().item() <= SCORE_TOL
    assert 5.0 < eer_ref < 35.0
    assert eer_lo - 1e-9 <= eer_got <= eer_hi + 1e-9
    assert abs(eer_got - eer_ref) <= 0.2


@pytest.mark.parametrize("dtype", ["fp32", "fp16x3"])
def test_benchmarked_student_4096_trials_eer_unchanged_to_2dp(afx_mod, dtype):
    """`north_star`: "EER unchanged to 2 d.p." -- met AS WRITTEN on the 4096-trial fixture of the benchmarked student by the
    modes that carry fp32 accuracy through the trunk: the build's EER equals the oracle's to two
    decimals (|delta| < 0.005 percentage points) and every score is within 1e-5 / 1e-3 of the oracle's."""
    engine, synth = afx_mod
    from conftest import load_golden
    from afx import harness
    z = load_golden("eer_student_4096.npz")
    n_batches = z["scores"].shape[0] // 64
    sd = synth.model_state_dict("ConformerModel", n_layers=6)
    eng = engine.Engine("conformer", n_layers=6, dtype=dtype)
    eng.load_state_dict(sd)
    got = torch.cat([eng.forward(synth.waveforms(64, 64000, batch_idx=9000 + i).cuda())[:, 1].cpu() for i in range(n_batches)])
    ref = torch.from_numpy(z["scores"])
    labels = z["labels"].astype(int)
    d = (got - ref).abs()
    eer_ref, eer_got = float(z["eer"]), harness.calculate_EER(got.numpy(), labels)
    print(f"{dtype}: {len(labels)} trials, max|dscore| {d.max().item():.2e}; EER oracle {eer_ref:.4f} % build {eer_got:.4f} %")
    assert d.max().item() <= (1e-5 if dtype == "fp32" else 5e-5)
    assert abs(eer_got - eer_ref) < 0.005 and f"{eer_got:.2f}" == f"{eer_ref:.2f}"


def test_length_policy_as_one_batched_device_op(afx_mod):
    """pad-by-tiling, first-N crop and random-start crop for a ragged batch in one
    kernel, against the reference policies restated in oracle/pre.py (data/test_set.py:139-248)."""
    import numpy as np
    from afx import harness
    from oracle import pre
    g = torch.Generator().manual_seed(3)
    lens = [1, 7, 3999, 16000, 16001, 40000]
    clips = [torch.randn(n, generator=g) for n in lens]
    D = 16000
    got = harness.batch_adjust_duration(clips, D).cpu()
    for b, c in enumerate(clips):
        assert torch.equal(got[b], pre.adjust_duration(c, D))
        assert np.array_equal(got[b].numpy(), pre.pad_tile(c.numpy(), D))
    starts = [0, 0, 0, 0, 1, 12345]
    got = harness.batch_adjust_duration(clips, D, starts=starts).cpu()
    for b, (c, s0) in enumerate(zip(clips, starts)):
        ref = pre.adjust_duration(c, D) if c.numel() < D else c[s0:s0 + D]
        assert torch.equal(got[b], ref)
    with pytest.raises(ValueError):
        harness.batch_adjust_duration(clips, D, starts=[0, 0, 0, 1, 0, 0])


def test_streaming_scores_equal_the_reference_on_every_window(afx_mod):
    """BASELINE config 5 semantics (afx/streaming.py): every hop, each stream's score is the model's
    score of the last `window` samples (history repeated while it is shorter) -- checked against the
    CPU oracle at every hop, across the ring wrap-around."""
    engine, synth = afx_mod
    from afx.streaming import SlidingWindowScorer
    from oracle import models, pre
    sd = synth.model_state_dict("ConformerModel", n_layers=1, n_encoders=1)
    eng = engine.Engine("conformer", n_layers=1, dtype="fp16", conf_blocks=1)
    eng.load_state_dict(sd)
    S, W, H = 3, 16000, 4000
    sc = SlidingWindowScorer(eng, S, window=W, hop=H)
    stream = synth.waveforms(S, 9 * H, batch_idx=77)
    for step in range(9):
        got = sc.push(stream[:, step * H:(step + 1) * H].cuda()).cpu()
        hist = stream[:, : (step + 1) * H]
        win = torch.stack([pre.adjust_duration(hist[s], W) if hist.shape[1] < W else hist[s, -W:] for s in range(S)])
        ref = models.conformer_forward(sd, win)[:, 1]
        assert (got - ref).abs().max().item() <= SCORE_TOL, f"hop {step}"
    with pytest.raises(ValueError):
        sc.push(torch.zeros(S, H))


@pytest.mark.parametrize("arch", ["conformer", "xlsr_aasist"])
def test_incremental_streaming_is_bit_identical_to_rescoring_the_window(afx_mod, arch):
    """BASELINE config 5 with exact reuse (afx/streaming.py IncrementalScorer): conv layers 0-5 are computed once per
    frame and cached, conv layer 6 and everything bidirectional is recomputed -- every hop's scores must EQUAL, bit for
    bit, the scores of the full forward on the same window (SlidingWindowScorer, itself checked against the oracle at
    every hop above), through the warm-up, the first full window and many hops of ring wrap-around."""
    engine, synth = afx_mod
    from afx.streaming import IncrementalScorer, SlidingWindowScorer
    from oracle import models
    if arch == "conformer":
        sd = synth.model_state_dict("ConformerModel", n_layers=1, n_encoders=1)
        eng = engine.Engine("conformer", n_layers=1, dtype="fp16", conf_blocks=1)
        ofwd = models.conformer_forward
    else:
        sd = synth.model_state_dict("XLSR_AASIST", n_layers=1)
        eng = engine.Engine("xlsr_aasist", n_layers=1, dtype="fp16")
        ofwd = models.xlsr_aasist_forward
    eng.load_state_dict(sd)
    S, W, H = 3, 16000, 4000
    inc = IncrementalScorer(eng, sd, S, window=W, hop=H)
    ref = SlidingWindowScorer(eng, S, window=W, hop=H)
    stream = synth.waveforms(S, 13 * H, batch_idx=91)
    for step in range(13):
        chunk = stream[:, step * H:(step + 1) * H].cuda()
        a = inc.push(chunk).clone()
        b = ref.push(chunk).clone()
        assert torch.equal(a, b), f"hop {step}: {(a - b).abs().max().item():.2e}"
    win = stream[:, -W:]
    assert (a.cpu() - ofwd(sd, win)[:, 1]).abs().max().item() <= SCORE_TOL
    with pytest.raises(ValueError, match="hop % 160"):
        IncrementalScorer(eng, sd, S, window=16000, hop=1000)


def test_distributed_scoring_over_rccl_with_one_rank(afx_mod, tmp_path):
    """afx.harness.produce_evaluation_file_distributed through the real RCCL backend (world size 1 on the
    one GPU of this box: same code path as N ranks -- shard, score, all-gather of (index, score) pairs,
    merge, write); the file must equal the single-process one."""
    import torch.distributed as dist
    engine, synth = afx_mod
    from afx import harness
    sd = synth.model_state_dict("ConformerModel", n_layers=1, n_encoders=1)
    eng = engine.Engine("conformer", n_layers=1, dtype="fp16", conf_blocks=1)
    eng.load_state_dict(sd)

    class Toy(torch.utils.data.Dataset):
        def __len__(self):
            return 7

        def __getitem__(self, i):
            return f"utt{i}", synth.waveforms(1, 16000, batch_idx=300 + i)[0], 0

    class Wrap(torch.nn.Module):
        def forward(self, x):
            return eng.forward(x)

    ref_names, ref_scores = harness.produce_evaluation_file(Toy(), Wrap(), "cuda", str(tmp_path / "a.txt"), batch_size=3, num_workers=0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29577", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        idx, sc = harness.produce_evaluation_file_distributed(Toy(), Wrap(), "cuda", str(tmp_path / "b.txt"), batch_size=3, num_workers=0)
    finally:
        dist.destroy_process_group()
    assert idx.tolist() == list(range(7))
    assert (tmp_path / "a.txt").read_text() == (tmp_path / "b.txt").read_text()
    assert [f"utt{i}" for i in range(7)] == ref_names and len(ref_scores) == 7


@pytest.mark.parametrize("pre_emphasis", [False, True])
def test_eval_loop_loss_and_accuracy_like_trainer_test(afx_mod, pre_emphasis):
    """afx.harness.evaluate == Trainer._test (trainer.py:85-132): pre-emphasis when configured (:104, Q7), model,
    weighted cross-entropy summed as loss x batch size, arg-max accuracy in percent, ragged last batch -- on a toy
    loader, against the same loop over the CPU oracle (scores within 1e-3 -> loss within 1e-3, accuracy identical
    because no toy sample sits on the decision boundary)."""
    engine, synth = afx_mod
    from afx import harness
    from oracle import models as omodels, pre
    sd = synth.model_state_dict("ConformerModel", n_layers=1, n_encoders=1)
    eng = engine.Engine("conformer", n_layers=1, dtype="fp16", conf_blocks=1)
    eng.load_state_dict(sd)

    class Wrap(torch.nn.Module):
        def forward(self, x):
            return eng.forward(x)

    n = 11
    waves = torch.cat([synth.waveforms(1, 16000, batch_idx=700 + i) for i in range(n)])
    ref_in = pre.pre_emphasis(waves) if pre_emphasis else waves
    ref_logits = omodels.conformer_forward(sd, ref_in)
    labels = (ref_logits[:, 1] > ref_logits[:, 0]).long()
    labels[::3] = 1 - labels[::3]  # a third of the labels wrong on purpose: accuracy is neither 0 nor 100
    assert (ref_logits[:, 1] - ref_logits[:, 0]).abs().min().item() > 5e-3  # nobody on the boundary
    loader = [([f"u{i}" for i in range(s, min(s + 4, n))], waves[s:s + 4], labels[s:s + 4].float().view(-1, 1))
              for s in range(0, n, 4)]  # batches of 4, 4, 3; labels arrive as float columns like the reference's
    loss_fn = torch.nn.CrossEntropyLoss(weight=torch.tensor([0.9, 0.1]).cuda())  # main.py:106,122
    prep = harness.PreEmphasis(enabled=pre_emphasis)
    loss, acc = harness.evaluate(Wrap(), loader, "cuda", loss_fn=loss_fn, preprocessor=prep)
    # trainer.py:85-132 over the oracle's logits
    ce = torch.nn.CrossEntropyLoss(weight=torch.tensor([0.9, 0.1]))
    want_loss = sum(ce(ref_logits[s:s + 4], labels[s:s + 4]).item() * len(labels[s:s + 4]) for s in range(0, n, 4)) / n
    want_acc = (ref_logits.max(dim=1)[1] == labels).sum().item() / n * 100
    assert abs(loss - want_loss) <= 1e-3 and acc == want_acc and 0 < acc < 100


def test_engine_is_bound_to_its_device_not_to_the_current_one(afx_mod):
    """The reference passes device=rank and never calls torch.cuda.set_device (main.py:48,78-82): an engine must keep
    weights, workspace and launches on ITS device and refuse tensors of another one instead of dereferencing them."""
    engine, synth = afx_mod
    from afx._lib import AfxError
    eng = engine.Engine("ssl", n_layers=1, dtype="fp16", device="cuda:0")
    assert eng.device == torch.device("cuda", 0)
    with pytest.raises(AfxError, match="lives on"):
        eng._on_device(_FakeOtherDevice(), "input")
    with pytest.raises(AfxError, match="GPU"):
        engine.Engine("ssl", n_layers=1, device="cpu")


def test_one_side_stream_and_one_copy_stream_per_gpu(afx_mod):
    """Hardware queues are a per-process resource shared among streams in order of first use (afx.engine.side_stream): every
    engine of a GPU runs its back-ends on THE side stream of that GPU and every scoring pass stages on THE copy stream -- no
    fresh stream per engine or per pass, none of them a priority stream, and both distinct from torch's current stream."""
    engine, synth = afx_mod
    from afx.harness import prefetch_to_device
    side, copy = engine.side_stream("cuda:0"), engine.side_stream(torch.device("cuda", 0), "copy")
    assert side is engine.side_stream(0) and copy is engine.side_stream("cuda:0", "copy")
    cur = torch.cuda.current_stream()
    assert len({side.cuda_stream, copy.cuda_stream, cur.cuda_stream}) == 3
    assert side.priority == 0 and copy.priority == 0
    sd = synth.model_state_dict("ConformerModel", n_layers=1, n_encoders=1)
    wave = synth.waveforms(2, 16000, batch_idx=7).cuda()
    outs = []
    for _ in range(2):  # two engines, one side stream
        eng = engine.Engine("conformer", n_layers=1, dtype="fp16", conf_blocks=1)
        eng.load_state_dict(sd)
        got = eng.forward_overlapped(wave)
        assert eng._side is side
        eng.join()
        assert torch.equal(got, eng.forward(wave))
        outs.append(got.clone())
    assert torch.equal(outs[0], outs[1])
    host = wave.cpu()
    for _ in range(3):  # three passes, one copy stream: the staged batches arrive intact
        for _meta, x in prefetch_to_device(((i, host) for i in range(3)), "cuda:0"):
            assert torch.equal(x.cpu(), host)


class _FakeOtherDevice:
    """Stands for a tensor on another GPU (this box has one): only what Engine._on_device looks at."""
    is_cuda = True
    device = torch.device("cuda", 1)


def test_kd_forward_hooks_by_module_path(afx_mod):
    """trainer.py:156-195,263-270: ForwardHookManager.add_hook(model, module_path, requires_input, requires_output) +
    pop_io_dict() on the drop-in modules (one native call, so the values come from the engine's taps) against the
    oracle's intermediates at the same paths."""
    engine, synth = afx_mod
    from afx.kd import ForwardHookManager
    from models.conformer_baseline import MyModel
    from oracle import models as omodels
    stu = MyModel(device="cuda", ssl_cpkt_path=None, num_layers=2, order="first", n_encoders=2).to("cuda").eval()
    sd = synth.model_state_dict("ConformerModel", n_layers=2, n_encoders=2)
    stu.load_state_dict(sd)
    mgr = ForwardHookManager("cuda:0")
    mgr.add_hook(stu, "ssl_model.model.encoder.layers.1", requires_input=True, requires_output=True)
    mgr.add_hook(stu, "conformer.encoder_blocks.0", requires_input=False, requires_output=True)
    mgr.add_hook(stu, "ssl_model", requires_input=False, requires_output=True)
    with pytest.raises(AttributeError):
        mgr.add_hook(stu, "no.such.module")
    with pytest.raises(ValueError, match="no native tap"):
        mgr.add_hook(stu, "first_bn")
    wave = synth.waveforms(2, 16000, batch_idx=41)
    with torch.no_grad():
        out = stu(wave.cuda())
    io = mgr.pop_io_dict()
    assert mgr.pop_io_dict() == {}  # popped
    taps = {}
    ref = omodels.conformer_forward(sd, wave, taps=taps)
    assert (out.cpu() - ref).abs().max().item() <= SCORE_TOL

    def rel(a, b):
        return ((a.cpu().reshape(-1) - b.reshape(-1)).norm() / b.norm()).item()
    lay = io["ssl_model.model.encoder.layers.1"]
    assert lay["input"].shape == (2, 49, 1024) and rel(lay["input"], taps["layer0"]) < 2e-3
    assert rel(lay["output"], taps["layer1"]) < 2e-3
    assert "input" not in io["conformer.encoder_blocks.0"]
    assert io["conformer.encoder_blocks.0"]["output"].shape == (2, 50, 144) and rel(io["conformer.encoder_blocks.0"]["output"], taps["block0"]) < 3e-3
    assert rel(io["ssl_model"]["output"], taps["ssl"]) < 2e-3
    mgr.clear()
    with torch.no_grad():
        stu(wave.cuda())
    assert mgr.pop_io_dict() == {}
    # clear() switched the engine's taps off again: the forward above kept no fp32 copies, so the forward is capturable
    # and a tap refreshed by it would have changed -- the "ssl" tap still holds the hooked forward's values
    wave2 = synth.waveforms(2, 16000, batch_idx=42)
    before = stu._afx_engine().tap("ssl").clone()
    with torch.no_grad():
        stu(wave2.cuda())
    assert torch.equal(stu._afx_engine().tap("ssl"), before)


def test_checkpoint_files_through_the_gpu_path(afx_mod, tmp_path):
    """SURVEY 8(f) row 3 end to end on the GPU: a fairseq-style ``xlsr2_300m.pt`` (config object of an absent package +
    'model' state_dict with pre-training heads) is what ``ssl_cpkt_path`` points at (models/fe.py:11-14), a fine-tuned
    checkpoint saved from a DDP-wrapped model ('module.' keys, main.py:176-179) is loaded the way main.py:98-103 /
    :391-395 do, and the scores equal the oracle's on the same tensors."""
    import sys
    import types
    engine, synth = afx_mod
    from afx.harness import f_state_dict_wrapper
    from models.xlsr_aasist import My_XLSR_AASIST
    from oracle import models as omodels
    # 1. an SSL checkpoint (all 24 layers, like xlsr2_300m.pt: My_XLSR_FE truncates AFTER loading, models/fe.py:63-74)
    #    whose values differ from the seeded defaults, so that loading it matters
    ssl = {k: v * 1.01 for k, v in synth.ssl_state_dict(24, prefix="").items()}
    fake = types.ModuleType("fairseq_absent_pkg2")

    class Cfg:
        pass
    Cfg.__module__, Cfg.__qualname__ = "fairseq_absent_pkg2", "Cfg"
    fake.Cfg = Cfg
    sys.modules["fairseq_absent_pkg2"] = fake
    try:
        path = tmp_path / "xlsr_like.pt"
        torch.save({"cfg": Cfg(), "model": dict(ssl, **{"mask_emb": torch.zeros(1024), "final_proj.weight": torch.zeros(768, 1024)})}, path)
    finally:
        del sys.modules["fairseq_absent_pkg2"]
    model = My_XLSR_AASIST(device="cuda", ssl_cpkt_path=str(path), num_layers=2, order="first").to("cuda").eval()
    got_ssl = {k: v.detach().cpu() for k, v in model.ssl_model.model.state_dict().items()}
    assert len(model.ssl_model.model.encoder.layers) == 2 and all(torch.equal(got_ssl[k], ssl[k]) for k in got_ssl)
    # 2. a fine-tuned checkpoint of the whole (truncated) model, saved with the DDP prefix
    full = {("ssl_model.model." + k): v for k, v in ssl.items() if k in got_ssl}
    full.update(synth.aasist_head_state_dict())
    ck = tmp_path / "finetuned_ep_1_acc_99.pt"
    torch.save({"module." + k: v for k, v in full.items()}, ck)
    wrapped = torch.nn.DataParallel(model)
    wrapped.load_state_dict(f_state_dict_wrapper(torch.load(ck, map_location="cuda"), data_parallel=True))
    model = wrapped.module
    wave = synth.waveforms(3, 16000, batch_idx=17)
    with torch.no_grad():
        got = model(wave.cuda()).cpu()
    ref = omodels.xlsr_aasist_forward(full, wave)
    assert (got - ref).abs().max().item() <= SCORE_TOL


@pytest.mark.parametrize("arch", ["conformer", "xlsr_aasist"])
def test_ragged_batch_scores_each_clip_as_if_alone(afx_mod, arch, tmp_path):
    """SURVEY 8(f) row 1, second half: clips of different lengths in ONE forward (afx_forward_ragged: key-padding masks
    in both attentions, zero padding past a clip's own frames in the positional / depthwise convs, per-length AASIST
    sub-batches).  Every clip's logits must equal what the same engine gives for that clip alone -- bit for bit: a
    masked key contributes exactly 0 -- and match the CPU oracle on the clip alone within the score tolerance."""
    engine, synth = afx_mod
    from afx import harness
    from oracle import models as omodels
    if arch == "conformer":
        sd = synth.model_state_dict("ConformerModel", n_layers=2, n_encoders=2)
        eng = engine.Engine("conformer", n_layers=2, dtype="fp16", conf_blocks=2)
        ofwd = omodels.conformer_forward
    else:
        sd = synth.model_state_dict("XLSR_AASIST", n_layers=2)
        eng = engine.Engine("xlsr_aasist", n_layers=2, dtype="fp16")
        ofwd = omodels.xlsr_aasist_forward
    eng.load_state_dict(sd)
    lens = [64000, 16000, 40321, 7000, 64000, 23456, 16000]  # 199, 49, 125, 21, 199, 72, 49 frames
    clips = [synth.waveforms(1, n, batch_idx=800 + i)[0] for i, n in enumerate(lens)]
    got = eng.forward_ragged([c.cuda() for c in clips]).cpu()
    for b, c in enumerate(clips):
        alone = eng.forward(c[None].cuda()).cpu()[0]
        assert torch.equal(got[b], alone), f"clip {b} ({lens[b]} samples): {(got[b] - alone).abs().max().item():.2e}"
        ref = ofwd(sd, c[None])[0]
        assert (got[b] - ref).abs().max().item() <= SCORE_TOL
    with pytest.raises(Exception, match="400"):
        eng.forward_ragged([clips[0].cuda(), torch.zeros(100).cuda()])

    # the un-cropped scoring loop on top of it
    class Toy(torch.utils.data.Dataset):
        def __len__(self):
            return len(clips)

        def __getitem__(self, i):
            return f"u{i}", clips[i], 0

    class Wrap(torch.nn.Module):
        def forward_ragged(self, cs):
            return eng.forward_ragged([c.cuda() for c in cs])
    names, scores = harness.produce_evaluation_file_ragged(Toy(), Wrap(), "cuda", str(tmp_path / "r.txt"), batch_size=3, num_workers=0)
    assert names == [f"u{i}" for i in range(len(clips))]
    assert torch.allclose(torch.tensor(scores), got[:, 1], atol=0, rtol=0)


@pytest.mark.parametrize("arch", ["conformer", "xlsr_aasist"])
def test_ragged_batch_in_split_precision(afx_mod, arch):
    """The ragged forward in dtype "fp16x3" (producers write the next product's hi / lo operand planes in place, key-padding
    masks in the split-precision attention): every clip equals itself alone bit for bit and the fp32 oracle to 1e-5."""
    engine, synth = afx_mod
    from oracle import models as omodels
    if arch == "conformer":
        sd = synth.model_state_dict("ConformerModel", n_layers=2, n_encoders=2)
        eng = engine.Engine("conformer", n_layers=2, dtype="fp16x3", conf_blocks=2)
        ofwd = omodels.conformer_forward
    else:
        sd = synth.model_state_dict("XLSR_AASIST", n_layers=2)
        eng = engine.Engine("xlsr_aasist", n_layers=2, dtype="fp16x3")
        ofwd = omodels.xlsr_aasist_forward
    eng.load_state_dict(sd)
    lens = [64000, 16000, 40321, 7000, 64000]
    clips = [synth.waveforms(1, n, batch_idx=840 + i)[0] for i, n in enumerate(lens)]
    got = eng.forward_ragged([c.cuda() for c in clips]).cpu()
    for b, c in enumerate(clips):
        assert torch.equal(got[b], eng.forward(c[None].cuda()).cpu()[0]), b
        assert (got[b] - ofwd(sd, c[None])[0]).abs().max().item() <= 1e-5


def test_ragged_bit_identity_holds_across_tile_families(afx_mod):
    """The same statement at a batch large enough that its dense products run on OTHER tile instances than a clip's alone
    (48 clips: QKV / FC1 / out-proj / FC2 on the 8-wave 256-wide tiles; one clip alone: 128 x 64 tiles) -- every tile instance
    accumulates a row's K in the same order, so a clip's logits still equal the logits it gets alone, bit for bit.  (A second
    summation order for some batch sizes -- e.g. a K-split tile, measured in round 3 -- would end this)
    The one condition: the batch's longest clip and the clip alone must select the same FORM of the trunk attention -- the
    one-pass kernel up to 224 frames (4.49 s), the key-blocked one beyond (running max / sum: another rounding).  A batch that
    holds a longer clip scores its short clips through the blocked form: equal to alone within fp32 rounding, not bit for bit."""
    engine, synth = afx_mod
    sd = synth.model_state_dict("XLSR_AASIST", n_layers=2)
    eng = engine.Engine("xlsr_aasist", n_layers=2, dtype="fp16")
    eng.load_state_dict(sd)
    lens = [64000 if i % 3 else 20000 + 907 * i for i in range(48)]  # 62 ... 199 frames
    assert max(lens) == 64000
    clips = [synth.waveforms(1, n, batch_idx=1800 + i)[0] for i, n in enumerate(lens)]
    got = eng.forward_ragged([c.cuda() for c in clips]).cpu()
    for b in (0, 1, 7, 20, 33, 47):
        alone = eng.forward(clips[b][None].cuda()).cpu()[0]
        assert torch.equal(got[b], alone), f"clip {b}: {(got[b] - alone).abs().max().item():.2e}"
    # and a uniform batch of 48 against the same clips in batches of 3
    wave = synth.waveforms(48, 32000, batch_idx=77).cuda()
    whole = eng.forward(wave)
    parts = torch.cat([eng.forward(wave[i:i + 3]) for i in range(0, 48, 3)])
    assert torch.equal(whole, parts)
    # one 5-s clip (249 frames) in the batch: every clip goes through the key-blocked attention
    mixed = [synth.waveforms(1, 80000, batch_idx=1900)[0]] + clips[:5]
    gm = eng.forward_ragged([c.cuda() for c in mixed]).cpu()
    for b in (1, 2, 3):
        alone = eng.forward(mixed[b][None].cuda()).cpu()[0]
        assert (gm[b] - alone).abs().max().item() <= 2e-6


def test_deep_tile_switch_is_per_engine_and_changes_no_bit(afx_mod):
    """"gemm_small_deep" (the deep form of the 128x64 tile for products with at most two tiles per CU -- the teacher's
    N = 1024 products at batch 16) is a switch of ONE engine: engine A with it off and engine B with it on, driven
    alternately in one process, each keep their own choice (their profilers say which tile class ran), and the logits are
    the same bits (same k order per row in both forms)."""
    engine, synth = afx_mod
    sd = synth.model_state_dict("XLSR_AASIST", n_layers=2)
    a = engine.Engine("xlsr_aasist", n_layers=2, dtype="fp16")
    b = engine.Engine("xlsr_aasist", n_layers=2, dtype="fp16")
    a.load_state_dict(sd)
    b.load_state_dict(sd)
    a.set("gemm_small_deep", 0)
    wave = synth.waveforms(16, 64000, batch_idx=9).cuda()
    outs = []
    for eng, deep in ((a, False), (b, True), (a, False)):
        eng.profile_begin()
        outs.append(eng.forward(wave).clone())
        prof = eng.profile_end()
        assert (prof["gemm_deep_kernel<128x64>"]["launches"] > 0) == deep, (deep, {k: v["launches"] for k, v in prof.items() if v["launches"]})
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_scoring_loop_runs_forward_hooks(afx_mod, tmp_path):
    """main.py:211 calls ``model(batch_x)``: nn.Module.__call__, which is what runs forward (pre-)hooks.  The scoring loop's
    two-stream form calls ``forward_overlapped`` directly -- so it is only taken for a model WITHOUT hooks; with a hook
    registered the loop goes through ``model(x)`` and the hook sees every batch."""
    engine, synth = afx_mod
    from afx import harness
    from models.conformer_baseline import MyModel
    sd = synth.model_state_dict("ConformerModel", n_layers=2, n_encoders=2)
    m = MyModel(device="cuda", ssl_cpkt_path=None, num_layers=2, order="first", n_encoders=2).to("cuda").eval()
    m.load_state_dict(sd)

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 6

        def __getitem__(self, i):
            return f"utt{i}", synth.waveforms(1, 16000, batch_idx=300 + i)[0], 1

    seen = []
    assert harness._may_overlap(m)
    h = m.register_forward_hook(lambda mod, inp, out: seen.append(tuple(out.shape)))
    assert not harness._may_overlap(m)
    names, hooked = harness.produce_evaluation_file(DS(), m, "cuda", str(tmp_path / "hooked.txt"), batch_size=2, num_workers=0)
    assert seen == [(2, 2)] * 3
    h.remove()
    assert harness._may_overlap(m)
    _names, plain = harness.produce_evaluation_file(DS(), m, "cuda", str(tmp_path / "plain.txt"), batch_size=2, num_workers=0)
    assert hooked == plain and len(seen) == 3  # same bits either way; no hook, no call


def test_ragged_ssl_features(afx_mod):
    engine, synth = afx_mod
    sd = synth.ssl_state_dict(2)
    eng = engine.Engine("ssl", n_layers=2, dtype="fp16")
    eng.load_state_dict(sd)
    clips = [synth.waveforms(1, n, batch_idx=820 + i)[0].cuda() for i, n in enumerate([48000, 9000, 48000, 30000])]
    feats, frames = eng.ssl_ragged(clips)
    assert frames == [149, 27, 149, 93] and feats.shape == (4, 149, 1024)
    for b, c in enumerate(clips):
        alone = eng.ssl(c[None])[0]
        assert torch.equal(feats[b, : frames[b]], alone)
        assert bool((feats[b, frames[b]:] == 0).all())


@pytest.mark.parametrize("dtype,tol", [("fp16", 2e-3), ("fp32", 2e-5), ("fp16x3", 2e-5)])
def test_group_norm_extractor_mode(afx_mod, dtype, tol):
    """The wav2vec2-base feature extractor `north_star` names beside XLS-R's (fairseq extractor_mode="default":
    bias-free convs, GroupNorm(512,512) = per-utterance-and-channel normalisation over time on layer 0 only, GELU;
    SURVEY 8a row 1a) against the oracle's mode="default", after the conv stack and at the trunk's output."""
    engine, synth = afx_mod
    from oracle import ssl_trunk
    sd = synth.ssl_state_dict(2, extractor_mode="group_norm")
    assert "ssl_model.model.feature_extractor.conv_layers.0.2.weight" in sd
    assert "ssl_model.model.feature_extractor.conv_layers.0.0.bias" not in sd
    wave = synth.waveforms(3, 32000, batch_idx=61) + 0.02  # a DC offset: the group norm must remove per-channel means
    taps = {}
    ref = ssl_trunk.ssl_forward({k[len(synth.SSL_PREFIX):]: v for k, v in sd.items()}, wave, mode="default", taps=taps)
    eng = engine.Engine("ssl", n_layers=2, dtype=dtype, extractor_mode="group_norm")
    eng.load_state_dict(sd)
    eng.enable_taps()
    got = eng.ssl(wave.cuda()).cpu()

    def rel(a, b):
        return ((a.reshape(-1) - b.reshape(-1)).norm() / b.norm()).item()
    assert rel(eng.tap("conv").cpu(), taps["conv"]) < tol
    assert got.shape == ref.shape and rel(got, ref) < tol
    with pytest.raises(Exception, match="group-norm|whole clip"):
        eng.ssl_ragged([wave[0].cuda(), wave[1, :20000].cuda()])
    with pytest.raises(Exception, match="missing weight"):
        engine.Engine("ssl", n_layers=2, dtype=dtype, extractor_mode="group_norm").load_state_dict(synth.ssl_state_dict(2))
