"""AASIST back-end parity on a real MI355X.  The expected values of the first four
tests are outputs of the REFERENCE'S OWN code (tests/golden/aasist_*.npz, see
make_golden.py); the last one uses the CPU oracle for the whole teacher model."""
import numpy as np
import pytest
import torch

from conftest import load_golden, sub_sd

pytestmark = pytest.mark.gpu

TOL = dict(rtol=2e-4, atol=5e-5)


def _c(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _gat_params(K, sd):
    sc, sh = K.bn_fold(sd["bn.weight"], sd["bn.bias"], sd["bn.running_mean"], sd["bn.running_var"])
    p = dict(att_w=sd["att_proj.weight"], att_b=sd["att_proj.bias"], att_vec=sd["att_weight"].reshape(-1),
             w1=sd["proj_with_att.weight"], b1=sd["proj_with_att.bias"], w2=sd["proj_without_att.weight"],
             b2=sd["proj_without_att.bias"], bn_scale=sc, bn_shift=sh)
    return {k: v.contiguous().cuda() for k, v in p.items()}


def _hgat_params(K, sd):
    sc, sh = K.bn_fold(sd["bn.weight"], sd["bn.bias"], sd["bn.running_mean"], sd["bn.running_var"])
    p = dict(t1w=sd["proj_type1.weight"], t1b=sd["proj_type1.bias"], t2w=sd["proj_type2.weight"],
             t2b=sd["proj_type2.bias"], att_w=sd["att_proj.weight"], att_b=sd["att_proj.bias"],
             attM_w=sd["att_projM.weight"], attM_b=sd["att_projM.bias"], v11=sd["att_weight11"].reshape(-1),
             v22=sd["att_weight22"].reshape(-1), v12=sd["att_weight12"].reshape(-1), vM=sd["att_weightM"].reshape(-1),
             w1=sd["proj_with_att.weight"], b1=sd["proj_with_att.bias"], w2=sd["proj_without_att.weight"],
             b2=sd["proj_without_att.bias"], w1M=sd["proj_with_attM.weight"], b1M=sd["proj_with_attM.bias"],
             w2M=sd["proj_without_attM.weight"], b2M=sd["proj_without_attM.bias"], bn_scale=sc, bn_shift=sh)
    return {k: v.contiguous().cuda() for k, v in p.items()}


@pytest.fixture(scope="module")
def K():
    from afx import kernels
    return kernels


def test_graph_attention_layer_vs_reference_module(K):
    z = load_golden("aasist_modules.npz")
    y = K.gat(_c(z["gat.x"]), _gat_params(K, sub_sd(z, "gat.")), 2.0)
    np.testing.assert_allclose(y.cpu().numpy(), z["gat.y"], **TOL)


@pytest.mark.parametrize("tag", ["h64", "h32"])
def test_heterogeneous_layer_vs_reference_module(K, tag):
    z = load_golden("aasist_modules.npz")
    p = _hgat_params(K, sub_sd(z, tag + "."))
    y1, y2, ym = K.hgat(_c(z[tag + ".x1"]), _c(z[tag + ".x2"]), p, 100.0, master=_c(z[tag + ".master"]).reshape(1, -1))
    np.testing.assert_allclose(y1.cpu().numpy(), z[tag + ".y1"], **TOL)
    np.testing.assert_allclose(y2.cpu().numpy(), z[tag + ".y2"], **TOL)
    np.testing.assert_allclose(ym.cpu().numpy(), z[tag + ".ym"], **TOL)
    n1, n2, nm = K.hgat(_c(z[tag + ".x1"]), _c(z[tag + ".x2"]), p, 100.0, master=None)  # mean-of-nodes master
    np.testing.assert_allclose(n1.cpu().numpy(), z[tag + ".n1"], **TOL)
    np.testing.assert_allclose(n2.cpu().numpy(), z[tag + ".n2"], **TOL)
    np.testing.assert_allclose(nm.cpu().numpy(), z[tag + ".nm"], **TOL)


def test_graph_pool_vs_reference_module_descending_order(K):
    z = load_golden("aasist_modules.npz")
    sd = sub_sd(z, "pool.")
    y = K.graph_pool(_c(z["pool.x"]), sd["proj.weight"].reshape(-1).cuda(), sd["proj.bias"].cuda(), 0.5)
    assert y.shape == (3, 21, 64)
    np.testing.assert_allclose(y.cpu().numpy(), z["pool.y"], **TOL)


@pytest.mark.parametrize("tag,filts,first", [("rb_first", [1, 32], True), ("rb_down", [32, 64], False), ("rb_same", [64, 64], False)])
def test_residual_block_vs_reference_module(tag, filts, first):
    """Residual_block.forward stand-alone (afx_k_resblock) against the reference's own module outputs
    (models/aasist_modules.py:340-397, incl. Q2: conv1 sees x, bn1 is dead)."""
    from models.aasist_modules import Residual_block
    z = load_golden("aasist_modules.npz")
    rb = Residual_block(nb_filts=filts, first=first).eval()
    missing, unexpected = rb.load_state_dict(sub_sd(z, tag + "."), strict=True)
    assert not missing and not unexpected
    rb = rb.cuda()
    with torch.no_grad():
        y = rb(_c(z[tag + ".x"]))
    assert y.shape == z[tag + ".y"].shape
    np.testing.assert_allclose(y.cpu().numpy(), z[tag + ".y"], **TOL)
    with pytest.raises(RuntimeError, match="inference-only"):
        rb.train()(_c(z[tag + ".x"]))


@pytest.mark.parametrize("tag,T", [("t199", 199), ("t49", 49), ("t201", 201)])
def test_backend_vs_reference_forward(tag, T):
    """feats -> logits through afx_head_forward against XLSR_AASIST.forward itself."""
    from afx import engine, synth
    z = load_golden("aasist_backend.npz")
    sd = dict(synth.ssl_state_dict(1))
    sd.update(sub_sd(z, ""))
    eng = engine.Engine("xlsr_aasist", n_layers=1, dtype="fp16")
    eng.load_state_dict(sd)
    eng.enable_taps()
    feats = _c(z[tag + ".feats"])
    got = eng.head(feats).cpu().numpy()
    B = feats.shape[0]
    np.testing.assert_allclose(eng.tap("e_S").cpu().numpy().reshape(B, 42, 64), z[tag + ".e_S"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(eng.tap("e_T").cpu().numpy().reshape(B, T // 3, 64), z[tag + ".e_T"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(eng.tap("hidden").cpu().numpy().reshape(B, 160), z[tag + ".hidden"], rtol=5e-4, atol=5e-4)
    np.testing.assert_allclose(got, z[tag + ".logits"], rtol=0, atol=1e-3)


def test_teacher_model_end_to_end():
    """XLSR_AASIST end to end (2-layer trunk to keep the CPU oracle quick), 16 clips of 4 s, the seeded "lively" head
    (matrices x 1.5: node scores spread instead of sitting within 1e-6 of each other).

    Contract (DESIGN.md "Numerics"), every part for EVERY utterance:
      1. the fp32 back-end is exact: on the engine's own SSL features it equals the oracle back-end to 1e-5;
      2. the fp16 trunk is within 2e-3 relative L2 of the fp32 trunk;
      3. wherever the reference model keeps its GraphPool decisions under that trunk rounding
         (conftest.teacher_conditioning: same node sequences), the logits are within the 1e-3 score tolerance.
    Where a near-tie of the reference's own top-k flips, (1) still holds -- the deviation IS the oracle's response to
    the perturbed features -- and no tolerance is claimed; the count is printed."""
    from afx import engine, synth
    from conftest import teacher_conditioning
    sd = synth.model_state_dict("XLSR_AASIST", n_layers=2, head_scale=1.5)
    wave = torch.cat([synth.waveforms(8, 64000, batch_idx=2), synth.waveforms(8, 64000, batch_idx=3)])
    eng = engine.Engine("xlsr_aasist", n_layers=2, dtype="fp16")
    eng.load_state_dict(sd)
    _ref, _got, rows = teacher_conditioning(sd, wave, eng)
    print("teacher (2-layer trunk) per utterance (same top-k, |dlogit|, back-end alone):",
          [(r["same_topk"], f"{r['dlogit']:.1e}", f"{r['backend']:.1e}") for r in rows])
    assert max(r["backend"] for r in rows) <= 1e-5
    assert max(r["feat_rel_l2"] for r in rows) < 2e-3
    kept = [r for r in rows if r["same_topk"]]
    assert kept, "no utterance kept its top-k decisions: nothing to compare"
    assert all(r["dlogit"] <= 1e-3 for r in kept), [r for r in kept if r["dlogit"] > 1e-3]


def test_dropin_models_package_teacher_and_student():
    """The reference-named classes: construct, load a reference-format state_dict (with
    the DDP 'module.' prefix handled as utils.py:13-43 does), eval forward on the GPU."""
    from afx import synth
    from afx.harness import f_state_dict_wrapper
    from models.conformer_baseline import MyModel
    from models.xlsr_aasist import My_XLSR_AASIST
    from oracle import models as omodels
    wave = synth.waveforms(3, 16000, batch_idx=5)
    # student: first-2 trunk + 2 Conformer blocks
    stu = MyModel(device="cuda", ssl_cpkt_path=None, num_layers=2, order="first", n_encoders=2).to("cuda").eval()
    sd = synth.model_state_dict("ConformerModel", n_layers=2, n_encoders=2)
    ck = {"module." + k: v for k, v in sd.items()}  # as saved from a DDP-wrapped model (main.py:176-179)
    stu.load_state_dict(f_state_dict_wrapper(ck, data_parallel=False))
    with torch.no_grad():
        got = stu(wave.cuda()).cpu()
        got3 = stu(wave.cuda().unsqueeze(-1)).cpu()
    ref = omodels.conformer_forward(sd, wave)
    assert (got - ref).abs().max().item() <= 1e-3
    assert torch.equal(got, got3)
    with pytest.raises(RuntimeError, match="inference-only"):
        stu.train()(wave.cuda())
    # teacher with a truncated trunk; weights are whatever the constructor drew (default
    # torch init for the head, like the reference) -> compare against the oracle on ITS state_dict
    torch.manual_seed(20240917)  # the constructor draws the head like the reference does: fix the draw
    tea = My_XLSR_AASIST(device="cuda", num_layers=1, order="last").to("cuda").eval()
    assert len(tea.ssl_model.model.encoder.layers) == 1 and tea.ssl_model.out_dim == 1024
    own = {k: v.detach().cpu() for k, v in tea.state_dict().items()}
    with torch.no_grad():
        feats = tea.ssl_model.extract_feat(wave.cuda()).cpu()
        got = tea(wave.cuda()).cpu()
    ssl, head = omodels.split(own)
    from oracle import aasist as oa, ssl_trunk as ot
    ref_feats = ot.ssl_forward(ssl, wave)
    assert ((feats - ref_feats).norm() / ref_feats.norm()).item() < 2e-3
    assert got.shape == (3, 2) and bool(torch.isfinite(got).all())
    exact = tea._afx_engine().head(ref_feats.cuda()).cpu()
    # default-init heads put GraphPool node scores within ~1e-6 of each other; fp32 summation order then
    # decides a near-tie on the odd utterance (seen: 2e-5 on one logit) -- hence 1e-4, not 1e-5, here;
    # the seeded-head fixtures above hold 1e-5
    assert (exact - oa.aasist_backend(head, ref_feats)).abs().max().item() <= 1e-4


def test_teacher_to_student_layer_copy_like_main_kd():
    """main_kd.py:127-141: ``student.load_state_dict(teacher.state_dict(), strict=False)`` followed by per-layer
    ``student...encoder.layers[i].load_state_dict(teacher...encoder.layers[j].state_dict())`` works on the drop-in
    modules as on the reference's, and the native engine picks the new weights up."""
    from afx import synth
    from models.conformer_baseline import MyModel
    from oracle import models as omodels
    torch.manual_seed(7)
    teacher = MyModel(device="cuda", ssl_cpkt_path=None, num_layers=3, order="first", n_encoders=1).to("cuda").eval()
    teacher.load_state_dict(synth.model_state_dict("ConformerModel", n_layers=3, n_encoders=1))
    student = MyModel(device="cuda", ssl_cpkt_path=None, num_layers=2, order="first", n_encoders=1).to("cuda").eval()
    wave = synth.waveforms(2, 16000, batch_idx=9)
    with torch.no_grad():
        before = student(wave.cuda()).cpu()
    res = student.load_state_dict(teacher.state_dict(), strict=False)
    assert not res.missing_keys and all("encoder.layers.2." in k for k in res.unexpected_keys)
    order = [2, 0]  # custom_order_copy_weights
    for i, j in enumerate(order):
        student.ssl_model.model.encoder.layers[i].load_state_dict(teacher.ssl_model.model.encoder.layers[j].state_dict(), strict=False)
    with torch.no_grad():
        got = student(wave.cuda()).cpu()
    assert (got - before).abs().max().item() > 1e-4  # the engine saw the new weights
    tsd = {k: v.detach().cpu() for k, v in teacher.state_dict().items()}
    want_sd = {k: v for k, v in tsd.items() if "encoder.layers." not in k}
    for i, j in enumerate(order):
        pre_t, pre_s = f"ssl_model.model.encoder.layers.{j}.", f"ssl_model.model.encoder.layers.{i}."
        want_sd.update({pre_s + k[len(pre_t):]: v for k, v in tsd.items() if k.startswith(pre_t)})
    ref = omodels.conformer_forward(want_sd, wave)
    assert (got - ref).abs().max().item() <= 1e-3


@pytest.mark.parametrize("dtype", ["fp16", "fp16x3", "fp32"])
@pytest.mark.parametrize("arch", ["xlsr_aasist", "conformer"])
def test_scoring_loop_overlaps_the_backend_with_the_next_trunk_bit_for_bit(tmp_path, arch, dtype):
    """main.py:199-221 scores batch after batch and reads the scores at the end; afx.harness.produce_evaluation_file issues
    the back-end of batch i (AASIST graph head / Conformer head) on a side stream under the trunk of batch i+1
    (afx_trunk_forward / afx_head_from_workspace, two workspaces alternating).  Same kernels on the same data: every score
    must equal the one-stream forward's bit for bit, over more batches than workspaces, a ragged last batch and changing
    batch sizes.  In split precision too: the trunk leaves its features for the Conformer head as a pair-form operand, and the
    head call -- a second entry into the library, on another stream -- has to know (round 4: it did not, and scored garbage
    while the one-stream forward was right; found by the overflow guard in bench.py's fp16x3 run)."""
    from afx import engine, harness, synth
    if arch == "xlsr_aasist":
        from models.xlsr_aasist import My_XLSR_AASIST as Cls
        sd = synth.model_state_dict("XLSR_AASIST", n_layers=2, head_scale=1.5)
        eng = engine.Engine("xlsr_aasist", n_layers=2, dtype=dtype)
        kw = {}
    else:
        from models.conformer_baseline import MyModel as Cls
        sd = synth.model_state_dict("ConformerModel", n_layers=2, n_encoders=2)
        eng = engine.Engine("conformer", n_layers=2, dtype=dtype, conf_blocks=2)
        kw = dict(n_encoders=2)
    eng.load_state_dict(sd)
    # (three times round the batch-size list, twice: 42 concurrent batches per combination.  Round 4 found the AASIST back-end in
    # fp16x3 NOT bit-stable here -- about one batch in three moved, profiles/r04_two_stream_race.txt -- while conv layer 0 of that
    # mode was the exact path's VALU kernel + a split launch; since it runs the matrix-core kernel, 0 of 105)
    waves = [synth.waveforms(b, 16000, batch_idx=700 + i).cuda() for i, b in enumerate([5, 5, 5, 3, 7, 5, 1] * 3)]
    assert eng.overlap_is_bit_stable
    want = [eng.forward(w).clone() for w in waves]
    for form in ("overlap", "lanes", "overlap"):  # the back-end beside the next trunk; whole forwards on alternating streams; and back
        eng.set_issue(form)
        for _ in range(2):  # twice: the second pass starts from slots that have a pending head / a busy side lane
            got = [eng.forward_overlapped(w) for w in waves]
            eng.join()
            for g, w_ in zip(got, want):
                assert torch.equal(g, w_), form
    assert torch.equal(eng.forward(waves[0]), want[0])  # and the one-stream call is unaffected afterwards
    assert eng.overlap_pays(waves[0]) in (True, False) and eng._issue in ("overlap", "lanes")  # the probe picks a form and says which
    got = [eng.forward_overlapped(w) for w in waves[:5]]
    eng.join()
    assert all(torch.equal(g, w_) for g, w_ in zip(got, want))
    eng.check_finite()

    class Toy(torch.utils.data.Dataset):
        def __len__(self):
            return 13

        def __getitem__(self, i):
            return f"utt{i}", synth.waveforms(1, 16000, batch_idx=900 + i)[0], 0
    model = Cls(device="cuda", ssl_cpkt_path=None, num_layers=2, order="first", **kw).to("cuda").eval()
    model.load_state_dict(sd)
    model.set_precision(dtype)
    names, scores = harness.produce_evaluation_file(Toy(), model, "cuda", str(tmp_path / "s.txt"), batch_size=4, num_workers=0)
    with torch.no_grad():
        ref = torch.cat([model(torch.stack([Toy()[i][1] for i in range(j, min(j + 4, 13))]).cuda())[:, 1] for j in range(0, 13, 4)]).cpu()
    assert names == [f"utt{i}" for i in range(13)]
    assert torch.equal(torch.tensor(scores, dtype=torch.float32), ref)
    assert (tmp_path / "s.txt").read_text().splitlines()[0].startswith("utt0 ")


@pytest.mark.parametrize("dtype,mode", [("fp32", "layer_norm"), ("fp16", "group_norm"), ("fp16x3", "group_norm")])
def test_valu_conv0_kernels_are_exact_beside_a_matrix_core_kernel(dtype, mode):
    """Round 4's two-stream defect, narrowed (profiles/r04_two_stream_race.txt): the VALU conv-layer-0 kernel, compiled with its
    10-tap loop in packed fp32 math, returned wrong low halves in lanes 48-63 of a frame whenever an fp16 GEMM kernel -- the vendor
    library's as well as this repository's -- started beside it (the vendor's GEMM moved 21 of 24 batches); compiled with the loop
    scalar it never moved.  The kernels that hold such a loop (exact mode's ``conv0_kernel<F32T>``, the group-norm extractor's
    ``conv0_gn_*``) are scalar there now (``scalar_only``) and this is the run-time guard: the trunk's features beside a stream of
    vendor fp16 GEMMs equal the features computed alone, bit for bit, 24 batches x 2.  (The instruction-level mechanism is not
    pinned down: tools/pk_hazard_probe.hip does not reproduce it in isolation.)"""
    from afx import engine, synth
    sd = synth.ssl_state_dict(1, extractor_mode=mode) if mode == "group_norm" else synth.ssl_state_dict(1)
    eng = engine.Engine("ssl", n_layers=1, dtype=dtype, extractor_mode=mode)
    eng.load_state_dict(sd)
    waves = [synth.waveforms(5, 16000, batch_idx=1200 + i).cuda() for i in range(24)]
    want = [eng.ssl(w).clone() for w in waves]
    a, b = (torch.randn(2048, 2048, device="cuda").half() for _ in range(2))
    c = torch.empty(2048, 2048, device="cuda", dtype=torch.float16)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    moved = []
    for rep in range(2):
        got = []
        for w in waves:
            with torch.cuda.stream(side):
                torch.mm(a, b, out=c)
                torch.mm(a, b, out=c)
            got.append(eng.ssl(w).clone())
        torch.cuda.synchronize()
        moved += [(rep, i) for i, (g, w_) in enumerate(zip(got, want)) if not torch.equal(g, w_)]
    assert not moved, f"{len(moved)} of 48 batches moved beside the GEMM stream: {moved[:8]}"


def test_packed_math_matrix_core_kernels_are_exact_beside_vendor_gemms():
    """The other side of round 4's build split: the GEMM tiles, the attention kernels and the fused Conformer
    chains keep their packed fp32 epilogue / softmax math.  They run it beside MFMAs by construction; this puts them beside the
    neighbours that moved 45-57 of 60 launches of the packed VALU conv0 kernel -- the vendor library's 2048^3 fp16 / bf16 GEMMs on a
    second stream -- and compares every output with the first one, bit for bit (tools/diag_pk_units.py is the long form:
    profiles/r04_pk_units_beside_vendor_gemm.txt, 0 of 40 in 45 cases)."""
    from afx import engine, kernels as K, synth
    g = torch.Generator(device="cuda").manual_seed(3)

    def rnd(*shape, dt=torch.float32, scale=1.0):
        return (torch.randn(*shape, generator=g, device="cuda") * scale).to(dt)
    a_big, w_fc1, b_fc1 = rnd(12736, 1024, dt=torch.float16), rnd(4096, 1024, dt=torch.float16, scale=0.03), rnd(4096)
    a_t, w_out, b_out, resid = rnd(3184, 1024, dt=torch.float16), rnd(1024, 1024, dt=torch.float16, scale=0.03), rnd(1024), rnd(3184, 1024)
    qkv = rnd(16 * 199, 3072, dt=torch.float16)
    eng = engine.Engine("conformer", n_layers=1, dtype="fp16", conf_blocks=2)
    eng.load_state_dict(synth.model_state_dict("ConformerModel", n_layers=1, n_encoders=2))
    feats = rnd(16, 199, 1024)
    victims = {"256-wide tile + GELU": lambda: K.gemm("fp16", a_big, w_fc1, bias=b_fc1, act="gelu", out_f=False, out_h=True)[1],
               "deep tile + residual": lambda: K.gemm("fp16", a_t, w_out, bias=b_out, resid=resid, out_f=True, out_h=False)[0],
               "attention": lambda: K.mhsa("fp16", qkv, 16, 199, 16),
               "Conformer head": lambda: eng.head(feats)}
    side = torch.cuda.Stream()
    moved = {}
    for dt in (torch.float16, torch.bfloat16):
        a, b = (torch.randn(2048, 2048, device="cuda").to(dt) for _ in range(2))
        c = torch.empty(2048, 2048, device="cuda", dtype=dt)
        torch.mm(a, b, out=c)
        for name, fn in victims.items():
            ref = fn().clone()
            torch.cuda.synchronize()
            for _ in range(12):
                with torch.cuda.stream(side):
                    torch.mm(a, b, out=c)
                    torch.mm(a, b, out=c)
                out = fn()
                torch.cuda.synchronize()
                if not torch.equal(out, ref):
                    moved[(name, str(dt))] = moved.get((name, str(dt)), 0) + 1
    assert not moved, moved
