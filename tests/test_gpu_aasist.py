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


def test_teacher_model_scores_match_oracle():
    """XLSR_AASIST end to end (2-layer trunk to keep the CPU oracle quick), B=5."""
    from afx import engine, synth
    from oracle import models
    sd = synth.model_state_dict("XLSR_AASIST", n_layers=2)
    wave = synth.waveforms(5, 64000, batch_idx=2)
    ref = models.xlsr_aasist_forward(sd, wave)
    eng = engine.Engine("xlsr_aasist", n_layers=2, dtype="fp16")
    eng.load_state_dict(sd)
    got = eng.forward(wave.cuda()).cpu()
    err = (got - ref).abs().max().item()
    assert err <= 1e-3, f"max |dlogit| {err:.3e}\n{got}\n{ref}"
