"""CPU-only tests: the C-ABI library loads and exports every symbol include/afx.h
declares (no compute calls without a GPU), the ctypes table matches the header, the
product fails loudly without a GPU, synthetic weights are deterministic and carry the
reference key names, and the N>1 score path works over gloo with world_size 2."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")


def header_symbols():
    src = open(os.path.join(ROOT, "include", "afx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(afx_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()  # hipcc cross-compiles gfx950 without a GPU
    from afx import _lib
    return _lib


def test_library_exports_every_symbol_of_the_header(built):
    names = header_symbols()
    assert len(names) >= 30
    lib = ctypes.CDLL(built.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_ctypes_table_covers_the_header(built):
    assert sorted(built.SIGNATURES) == header_symbols()
    built.lib()
    assert built.lib().afx_version().startswith(b"afx")
    assert built.lib().afx_num_frames(64000) == 199
    assert built.lib().afx_num_frames(16000) == 49
    assert built.lib().afx_num_frames(64600) == 201
    assert built.lib().afx_num_frames(300) == 0


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_fails_loudly_without_a_gpu(built):
    from afx import engine
    from afx._lib import AfxError
    with pytest.raises(AfxError, match="no CPU fallback"):
        engine.Engine("ssl", n_layers=1)
    cfg = built.Config(0, 1, 1, 144, 4, 31, 4, 0, 0.97)
    h = ctypes.c_void_p()
    assert built.lib().afx_create(ctypes.byref(cfg), ctypes.byref(h)) != 0
    assert b"no HIP device" in built.lib().afx_last_error()


def test_product_package_never_imports_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M):
                    bad.append(f)
    assert not bad, bad


def test_synthetic_weights_are_deterministic_and_reference_named():
    from afx import synth
    a = synth.model_state_dict("ConformerModel", n_layers=1)
    b = synth.model_state_dict("ConformerModel", n_layers=1)
    assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)
    for k in ("ssl_model.model.feature_extractor.conv_layers.0.0.weight",
              "ssl_model.model.feature_extractor.conv_layers.6.2.1.bias",
              "ssl_model.model.encoder.pos_conv.0.weight_g", "ssl_model.model.encoder.layers.0.self_attn.q_proj.weight",
              "ssl_model.model.encoder.layer_norm.weight", "LL.weight", "first_bn.running_var",
              "conformer.class_token", "conformer.encoder_blocks.3.attn.fn.rel_pos_emb.weight",
              "conformer.encoder_blocks.0.conv.net.4.conv.weight", "conformer.fc5.bias"):
        assert k in a, k
    assert a["conformer.encoder_blocks.0.conv.net.4.conv.weight"].shape == (288, 1, 31)
    t = synth.model_state_dict("XLSR_AASIST", n_layers=1)
    n_head = sum(v.numel() for k, v in t.items() if not k.startswith("ssl_model.") and v.dtype.is_floating_point
                 and "running" not in k)
    assert n_head == 447242  # back-end parameter count
    assert torch.equal(synth.waveforms(2, 100), synth.waveforms(2, 100))
    with pytest.raises(ValueError, match="not found"):
        synth.model_state_dict("NoSuchModel")


def test_shard_and_merge_roundtrip_single_process():
    from afx import dist as adist
    n, W = 11, 4
    shards = [adist.shard_indices(n, r, W) for r in range(W)]
    assert all(s.numel() == 3 for s in shards)
    idx = torch.cat(shards)
    scores = torch.where(idx >= 0, idx.float() * 0.5, torch.full_like(idx, 99.0, dtype=torch.float32))
    i2, s2 = adist.merge_scores(idx, scores)
    assert i2.tolist() == list(range(n))
    assert s2.tolist() == [0.5 * i for i in range(n)]


WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from afx import dist as adist
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n = 7
idx = adist.shard_indices(n, rank, world)
scores = torch.where(idx >= 0, idx.float() * 1.25 - 3.0, torch.zeros(idx.numel()))
gi, gs = adist.all_gather_scores(idx.to(torch.int32), scores, world)
mi, ms = adist.merge_scores(gi, gs)
assert mi.tolist() == list(range(n)), mi
assert ms.tolist() == [i * 1.25 - 3.0 for i in range(n)], ms
if rank == 0:
    print("GATHER_OK", mi.tolist())
dist.destroy_process_group()
"""


def test_score_all_gather_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29731", str(script), PKG],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "GATHER_OK [0, 1, 2, 3, 4, 5, 6]" in r.stdout


def test_dropin_models_package_has_reference_names_and_keys():
    """Class names resolved by main.py's globals() (main.py:20-21, main_kd.py:22), ctor
    kwargs, state_dict keys and the ValueErrors of models/fe.py:60-62,81-87."""
    from afx import synth
    import models  # noqa: F401
    from models.conformer_baseline import Model as ConformerModel, MyModel as MyConformerModel  # noqa: F401
    from models.models import SSLModel  # noqa: F401
    from models.xlsr_aasist import XLSR_AASIST, My_XLSR_AASIST  # noqa: F401
    from models.aasist_modules import GraphAttentionLayer, GraphPool, HtrgGraphAttentionLayer, Residual_block  # noqa: F401
    from models.fe import middle_indices
    assert middle_indices(24, 4) == [10, 11, 12, 13]
    stu = MyConformerModel(device="cpu", ssl_cpkt_path=None, num_layers=6, order="first")
    assert set(stu.state_dict()) == set(synth.model_state_dict("ConformerModel", n_layers=6))
    assert len(stu.ssl_model.model.encoder.layers) == 6 and stu.ssl_model.out_dim == 1024
    tea = My_XLSR_AASIST(device="cpu", num_layers=3, order="middle")
    assert set(tea.state_dict()) == set(synth.model_state_dict("XLSR_AASIST", n_layers=3))
    n_head = sum(p.numel() for n, p in tea.named_parameters() if not n.startswith("ssl_model."))
    assert n_head == 447242
    for bad in (dict(num_layers=0), dict(num_layers=25), dict(num_layers=2, order="custom"),
                dict(num_layers=2, order="custom", custom_order=(1, 2))):
        with pytest.raises(ValueError):
            My_XLSR_AASIST(device="cpu", **bad)
    cus = My_XLSR_AASIST(device="cpu", num_layers=2, order="custom", custom_order=[5, 1])
    assert len(cus.ssl_model.model.encoder.layers) == 2
    with pytest.raises(RuntimeError, match="inference-only"):
        tea(torch.zeros(1, 16000))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tea.eval()(torch.zeros(1, 16000))


def test_harness_helpers_match_reference_semantics(tmp_path):
    import numpy as np
    from afx import harness
    sd = {"a.w": 1, "module.b": 2}
    assert list(harness.f_state_dict_wrapper(sd, data_parallel=True)) == ["module.a.w", "module.b"]
    assert list(harness.f_state_dict_wrapper(sd, data_parallel=False)) == ["a.w", "b"]
    z = np.load(os.path.join(ROOT, "tests", "golden", "pre_eer.npz"))
    assert abs(harness.calculate_EER(z["scores"], z["labels"]) - float(z["eer"])) < 1e-9
    assert harness.adjust_duration(torch.from_numpy(z["short"]), 24).tolist() == z["tiled"].tolist()
    p = tmp_path / "sub" / "scores.txt"
    harness.write_score_file(str(p), ["u1", "u2"], [-2.9056100845336914, 0.5])
    assert p.read_text() == "u1 -2.9056100845336914\nu2 0.5\n"  # format of results/**.txt


def test_fairseq_style_checkpoint_loads_without_fairseq(tmp_path):
    """an ``xlsr2_300m.pt``-style file -- {"cfg": <object of an absent package>, "model":
    state_dict with pre-training heads} -- is read without fairseq: config objects are stubbed while
    unpickling, the unused heads are dropped, extra encoder layers beyond the trunk's are ignored."""
    import sys
    import types
    from afx import host, synth
    sd = synth.ssl_state_dict(2, prefix="")
    fake = types.ModuleType("fairseq_absent_pkg")

    class Wav2Vec2Config:  # stands for fairseq.models.wav2vec.wav2vec2.Wav2Vec2Config
        def __init__(self):
            self.encoder_layers = 2
    Wav2Vec2Config.__module__ = "fairseq_absent_pkg"
    Wav2Vec2Config.__qualname__ = "Wav2Vec2Config"
    fake.Wav2Vec2Config = Wav2Vec2Config
    sys.modules["fairseq_absent_pkg"] = fake
    try:
        model = dict(sd)
        model["mask_emb"] = torch.zeros(1024)
        model["quantizer.vars"] = torch.zeros(1, 640, 384)
        model["project_q.weight"] = torch.zeros(768, 768)
        model["final_proj.weight"] = torch.zeros(768, 1024)
        path = tmp_path / "xlsr_like.pt"
        torch.save({"cfg": Wav2Vec2Config(), "args": None, "model": model}, path)
    finally:
        del sys.modules["fairseq_absent_pkg"]  # from here on the class cannot be imported, as on a box without fairseq
    trunk = host.Wav2Vec2Trunk(n_layers=1)
    with torch.no_grad():
        for p in trunk.parameters():
            p.zero_()
    host.load_ssl_checkpoint(trunk, str(path))
    own = trunk.state_dict()
    assert all(torch.equal(own[k], sd[k]) for k in own)  # layer 0 of the 2-layer file, heads ignored
    torch.save({"model": {k: v for k, v in sd.items() if "fc2" not in k}}, path)
    with pytest.raises(KeyError, match="lacks"):
        host.load_ssl_checkpoint(trunk, str(path))


DIST_SCORE_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from afx import harness
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
calls = []

class Toy(torch.utils.data.Dataset):
    def __len__(self):
        return 9
    def __getitem__(self, i):
        calls.append(i)  # the reference datasets decode an audio file here (data/test_set.py)
        return f"clip_{i:03d}", torch.full((8,), float(i)), 0

class Model(torch.nn.Module):
    def forward(self, x):
        return torch.stack([-x[:, 0], x[:, 0] * 0.5], dim=1)

out = sys.argv[2]
mi, ms = harness.produce_evaluation_file_distributed(Toy(), Model(), "cpu", out, batch_size=2, num_workers=0)
assert mi.tolist() == list(range(9)) and ms.tolist() == [i * 0.5 for i in range(9)]
assert sorted(calls) == list(range(rank, 9, world)), calls  # every clip decoded once, by its own rank only
dist.barrier()
if rank == 0:
    lines = open(out).read().split("\n")
    assert lines[:9] == [f"clip_{i:03d} {i * 0.5}" for i in range(9)], lines
    print("DIST_SCORE_OK")
dist.destroy_process_group()
"""


def test_distributed_scoring_world_size_2_never_rereads_the_dataset(tmp_path):
    """produce_evaluation_file_distributed on two gloo ranks: utterances sharded r, r+W, ..., the ids travel with the
    shards (all_gather_object) -- rank 0 writes the file in dataset order without ever calling dataset[i] for a name."""
    script = tmp_path / "worker.py"
    script.write_text(DIST_SCORE_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29733", str(script), PKG, str(tmp_path / "scores.txt")],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "DIST_SCORE_OK" in r.stdout


def test_checkpoint_loader_does_not_execute_pickled_callables(tmp_path):
    """The fairseq-free checkpoint reader resolves only an allow-list of globals; anything else in the pickle --
    here os.system via __reduce__ -- becomes an inert stand-in instead of running."""
    import os as _os
    from afx import host, synth
    marker = tmp_path / "pwned"

    class Evil:
        def __reduce__(self):
            return (_os.system, (f"touch {marker}",))
    sd = synth.ssl_state_dict(1, prefix="")
    path = tmp_path / "evil.pt"
    torch.save({"cfg": Evil(), "model": sd}, path)
    trunk = host.Wav2Vec2Trunk(n_layers=1)
    host.load_ssl_checkpoint(trunk, str(path))
    assert not marker.exists()
    own = trunk.state_dict()
    assert all(torch.equal(own[k], sd[k]) for k in own)


@pytest.mark.parametrize("module,name", [
    ("torch.serialization", "os.system"),            # dotted name: pickle protocol 4 walks attributes
    ("torch.storage", "io.open"),
    ("torch._utils", "_import_dotted_name"),         # a real helper of an allowed module that imports anything
    ("torch.storage", "_load_from_bytes"),           # torch.load with the default (unrestricted) pickle
    ("posix", "system"), ("builtins", "eval"), ("builtins", "getattr"),
])
def test_checkpoint_loader_allow_list_has_no_attribute_walk(tmp_path, module, name):
    """the allow-list is exact (module, name) pairs resolved with one dict lookup on the module;
    a crafted 40-byte file that names a callable reachable THROUGH an allowed module must not run it."""
    from afx import host
    marker = tmp_path / "pwned"
    arg = f"touch {marker}" if "system" in name else ("open('%s','w')" % marker if name == "eval" else str(marker))

    def u(s):
        b = s.encode()
        return b"\x8c" + bytes([len(b)]) + b
    payload = b"\x80\x04" + u(module) + u(name) + b"\x93" + u(arg) + b"\x85R."  # STACK_GLOBAL, TUPLE1, REDUCE, STOP
    assert host._StubPickle._resolve(module, name) is None
    path = tmp_path / "crafted.pt"
    path.write_bytes(payload)
    with pytest.raises(Exception):
        host.load_ssl_checkpoint(host.Wav2Vec2Trunk(n_layers=1), str(path))
    assert not marker.exists()
    # what a real checkpoint needs still resolves to the real objects
    import collections
    assert host._StubPickle._resolve("collections", "OrderedDict") is collections.OrderedDict
    assert host._StubPickle._resolve("torch._utils", "_rebuild_tensor_v2") is torch._utils._rebuild_tensor_v2
    assert host._StubPickle._resolve("torch", "float16") is torch.float16
    assert host._StubPickle._resolve("torch", "FloatStorage") is torch.FloatStorage


def test_track_routing_and_checkpoint_sweep_like_main(tmp_path):
    """main.py:258-371,406-451: track loop (skip existing score files, comment suffix, InTheWild path derived from
    DF21's) and the --score_all_folder_path sweep (every *.pt, 'module.'-prefixed or not, comment from the file name)."""
    from afx import harness

    class Toy(torch.utils.data.Dataset):
        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

        def __getitem__(self, i):
            return f"u{i}", torch.full((4,), float(i)), 0

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.ones(1))

        def forward(self, x):
            return torch.stack([-x[:, 0], x[:, 0] * self.w], dim=1)

    built = []
    datasets = {"LA19": lambda: built.append("LA19") or Toy(3), "DF21": lambda: built.append("DF21") or Toy(2),
                "InTheWild": lambda: built.append("ITW") or Toy(1)}
    paths = {"LA19": str(tmp_path / "LA19.txt"), "DF21": str(tmp_path / "DF21.txt")}
    (tmp_path / "DF21_c1.txt").write_text("already there\n")
    done = harness.score_tracks(Model(), ["LA19", "DF21", "InTheWild"], datasets, paths, "cpu", 2, comment="c1", num_workers=0, log=lambda *_: None)
    assert built == ["LA19", "ITW"] and set(done) == {"LA19", "InTheWild"}  # DF21 skipped: its file existed
    assert (tmp_path / "LA19_c1.txt").read_text() == "u0 0.0\nu1 1.0\nu2 2.0\n"
    assert done["InTheWild"].endswith("InTheWild_c1.txt")
    with pytest.raises(ValueError, match="not found"):
        harness.score_tracks(Model(), ["XX"], datasets, paths, "cpu", 2, num_workers=0, log=lambda *_: None)
    ck = tmp_path / "ckpts"
    ck.mkdir()
    torch.save({"module.w": torch.tensor([2.0])}, ck / "run_ep_3_acc_97.pt")   # saved from a DDP-wrapped model (main.py:176-179)
    torch.save({"w": torch.tensor([3.0])}, ck / "run_ep_4_acc_98.pt")
    (ck / "notes.md").write_text("not a checkpoint")
    res = harness.score_all_checkpoints(str(ck), Model, ["LA19"], datasets, {"LA19": str(tmp_path / "s" / "LA19.txt")}, "cpu", 2,
                                        num_workers=0, log=lambda *_: None)
    assert len(res) == 2
    assert (tmp_path / "s" / "LA19_3_acc_97.pt.txt").read_text().split("\n")[1] == "u1 2.0"
    assert (tmp_path / "s" / "LA19_4_acc_98.pt.txt").read_text().split("\n")[1] == "u1 3.0"


def test_built_library_holds_no_in_place_cross_half_packed_fp32_op():
    """tools/scan_pk_hazard.py on the library this suite loads: no kernel may contain a packed fp32 op that overwrites the register
    pair whose HIGH register its LOW result reads -- the form that sat where round 4's two-stream defect showed (a conservative guard:
    DESIGN.md section 7 says what is and is not established about it)."""
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "real-time-deepfake-speech-detection_amd", "lib", "libafx.so")
    if not (os.path.exists(lib) and shutil.which("objcopy") and os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump")):
        pytest.skip("needs the built library, objcopy and the ROCm llvm-objdump")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "scan_pk_hazard.py"), lib], capture_output=True, text=True, timeout=600)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "refused): 0 in 0 kernels" in out.stdout
