"""Driver run UNDER AddressSanitizer + UBSan (LD_PRELOAD of the sanitizer runtime) by tests/test_cpu_asan.py: walks the
host side of libafx_asan.so -- the product sources compiled host-only against tests/asan/hip_host_shim.cpp (device memory =
malloc, launches = no-ops) -- through the C ABI with numpy arrays standing in for device tensors.  No torch in this process.
Usage: drive_host.py <libafx_asan.so> <weights.npz of {arch}/{key} arrays>.  Prints ASAN_DRIVE_OK at the end."""
import ctypes as C
import sys

import numpy as np

import os

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "real-time-deepfake-speech-detection_amd"))
from afx._lib import SIGNATURES  # noqa: E402  (the ctypes table of include/afx.h; the module itself needs no torch)

lib = C.CDLL(sys.argv[1])
for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)
    _fn.restype, _fn.argtypes = _res, _args
W = np.load(sys.argv[2])
P, I, L, Z, F = C.c_void_p, C.c_int, C.c_long, C.c_size_t, C.c_float


class Config(C.Structure):
    _fields_ = [("arch", I), ("dtype", I), ("n_layers", I), ("conf_emb", I), ("conf_heads", I), ("conf_kernel", I),
                ("conf_blocks", I), ("pre_emphasis", I), ("pre_emphasis_coef", F), ("extractor_mode", I)]


lib.afx_shim_launch_count.restype = L


def ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def ok(rc, what):
    assert rc == 0, f"{what}: {lib.afx_last_error().decode()}"


def fails(rc, needle):
    msg = lib.afx_last_error().decode()
    assert rc != 0 and needle in msg, (rc, msg, needle)


def create(arch, dtype=1, n_layers=1, blocks=1, kernel=31, mode=0):
    cfg = Config(arch, dtype, n_layers, 144, 4, kernel, blocks, 0, 0.97, mode)
    h = P()
    ok(lib.afx_create(C.cast(C.byref(cfg), SIGNATURES["afx_create"][1][0]), C.byref(h)), "afx_create")
    return h


def load(h, prefix, skip=()):
    for k in W.files:
        if not k.startswith(prefix + "/"):
            continue
        name = k[len(prefix) + 1:]
        if any(name.startswith(s) for s in skip):
            continue
        a = np.ascontiguousarray(W[k], dtype=np.float32)
        shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
        ok(lib.afx_load_weight(h, name.encode(), ptr(a), shape, a.ndim, None), "load " + name)


_alive = []


def ws_for(nbytes):
    """A fresh 'device' workspace; kept alive until the end of the run (ctypes holds only its address)."""
    a = np.empty(int(nbytes), dtype=np.uint8)
    _alive.append(a)
    if len(_alive) > 8:
        del _alive[0]
    return a


B, Ls = 3, 16000
wave = (0.1 * np.random.default_rng(0).standard_normal((B, Ls))).astype(np.float32)
logits = np.zeros((B, 2), np.float32)
SSL, AAS, CONF, HEAD = 0, 1, 2, 3

for arch, prefix in ((AAS, "xlsr_aasist"), (CONF, "conformer"), (SSL, "conformer")):
    for dtype in (1, 0, 2, 3):  # fp16, bf16, fp32 (exact mode walks other launch branches), fp16x3 (split precision)
        h = create(arch, dtype=dtype)
        fails(lib.afx_forward(h, ptr(wave), B, Ls, ptr(logits), ptr(ws_for(16)), 16, None), "not finalized")
        load(h, prefix)
        ok(lib.afx_finalize(h, None), "finalize")
        T = lib.afx_num_frames(Ls)
        if arch != SSL:
            n = lib.afx_workspace_bytes(h, B, Ls)
            ws = ws_for(n)
            ok(lib.afx_forward(h, ptr(wave), B, Ls, ptr(logits), ptr(ws), n, None), "forward")
            fails(lib.afx_forward(h, ptr(wave), B, Ls, ptr(logits), ptr(ws), n - 4096, None), "workspace too small")
            fails(lib.afx_forward(h, ptr(wave), B, 300, ptr(logits), ptr(ws), n, None), "too few")
            # the forward in two calls (back-end of one batch under the next batch's trunk: two workspaces)
            ws2 = ws_for(n)
            ok(lib.afx_trunk_forward(h, ptr(wave), B, Ls, ptr(ws2), n, None), "trunk forward")
            ok(lib.afx_head_from_workspace(h, B, Ls, ptr(logits), ptr(ws2), n, None), "head from workspace")
            fails(lib.afx_head_from_workspace(h, B, Ls, ptr(logits), ptr(ws2), n - 4096, None), "workspace too small")
            # taps: engine-owned fp32 copies of intermediates (memcpy / conversion launches inside the workspace bounds)
            ok(lib.afx_enable_taps(h, 1), "taps on")
            ok(lib.afx_forward(h, ptr(wave), B, Ls, ptr(logits), ptr(ws), n, None), "forward with taps")
            cnt = Z(0)
            ok(lib.afx_tap(h, b"ssl", None, 0, C.byref(cnt), None), "tap size")
            assert cnt.value == B * T * 1024
            out = np.empty(cnt.value, np.float32)
            ok(lib.afx_tap(h, b"ssl", ptr(out), cnt.value, C.byref(cnt), None), "tap copy")
            fails(lib.afx_tap(h, b"ssl", ptr(out), 8, C.byref(cnt), None), "too small")
            fails(lib.afx_tap(h, b"nope", None, 0, C.byref(cnt), None), "no tap named")
            ok(lib.afx_enable_taps(h, 0), "taps off")
            ok(lib.afx_check_finite(h, None), "overflow guard (the shim's device memory is zero-filled: nothing counted)")
            ok(lib.afx_engine_set(h, b"gemm_small_deep", 0), "per-engine tile switch")
            ok(lib.afx_forward(h, ptr(wave), B, Ls, ptr(logits), ptr(ws), n, None), "forward without the deep tile")
            ok(lib.afx_engine_set(h, b"gemm_small_deep", 1), "per-engine tile switch back")
            # per-class timing
            ok(lib.afx_profile_begin(h), "profile begin")
            ok(lib.afx_forward(h, ptr(wave), B, Ls, ptr(logits), ptr(ws), n, None), "profiled forward")
            nc = lib.afx_profile_num_classes()
            ms, fl, la = (C.c_double * nc)(), (C.c_double * nc)(), (C.c_longlong * nc)()
            ok(lib.afx_profile_end(h, nc, ms, fl, la), "profile end")
            assert sum(la) > 10
            # ragged batch: per-clip lengths, key-padding, AASIST buckets per distinct length
            if dtype != 2 or True:
                lens = (I * B)(16000, 9000, 16000)
                nr = lib.afx_ragged_workspace_bytes(h, B, Ls)
                wr = ws_for(nr)
                ok(lib.afx_forward_ragged(h, ptr(wave), B, Ls, lens, ptr(logits), ptr(wr), nr, None), "ragged forward")
                lens_bad = (I * B)(16000, 17000, 16000)
                fails(lib.afx_forward_ragged(h, ptr(wave), B, Ls, lens_bad, ptr(logits), ptr(wr), nr, None), "the batch rows hold")
                lens_short = (I * B)(16000, 100, 16000)
                fails(lib.afx_forward_ragged(h, ptr(wave), B, Ls, lens_short, ptr(logits), ptr(wr), nr, None), "too few")
            # head alone and the streaming tail
            feats = np.zeros((B, T, 1024), np.float32)
            nh = lib.afx_head_workspace_bytes(h, B, T)
            ok(lib.afx_head_forward(h, ptr(feats), B, T, ptr(logits), ptr(ws_for(nh)), nh, None), "head forward")
            T5 = 99
            c5 = np.zeros((B, 2 * T5, 512), np.float16 if dtype < 2 else np.float32)
            nt = lib.afx_tail_workspace_bytes(h, B, T5)
            ok(lib.afx_tail_forward(h, ptr(c5), B, T5, ptr(logits), ptr(ws_for(nt)), nt, None), "tail forward")
            if dtype != 3:
                ok(lib.afx_tail_forward_strided(h, ptr(c5), 2 * T5 * 512, B, T5, ptr(logits), ptr(ws_for(nt)), nt, None), "strided tail")
            else:
                fails(lib.afx_tail_forward_strided(h, ptr(c5), 2 * T5 * 512, B, T5, ptr(logits), ptr(ws_for(nt)), nt, None), "packed window")
            fails(lib.afx_tail_forward_strided(h, ptr(c5), 8, B, T5, ptr(logits), ptr(ws_for(nt)), nt, None), "shorter than a window")
            if dtype in (0, 1):  # KV-cached streaming mode: state object, 20 chunks (the 16-group ring wraps, the window fills)
                kv = P()
                ok(lib.afx_kv_create(h, B, C.byref(kv)), "kv create")
                assert lib.afx_kv_state_bytes(kv) > 0
                for hop in range(20):
                    nfr = 12 + (hop % 2)
                    f6 = np.zeros((B, nfr, 512), np.float32)
                    nk = lib.afx_kv_workspace_bytes(kv, nfr)
                    ok(lib.afx_kv_step(kv, ptr(f6), nfr, ptr(logits), ptr(ws_for(nk)), nk, None), "kv step")
                f6 = np.zeros((B, 17, 512), np.float32)
                fails(lib.afx_kv_step(kv, ptr(f6), 17, ptr(logits), ptr(ws_for(nk)), nk, None), "1..16 frames")
                fails(lib.afx_kv_step(kv, ptr(f6), 12, ptr(logits), ptr(ws_for(64)), 64, None), "workspace too small")
                lib.afx_kv_destroy(kv)
            elif dtype == 2:
                kv = P()
                fails(lib.afx_kv_create(h, B, C.byref(kv)), "half-precision")
            if arch == CONF:
                tok = np.zeros((B, T, 144), np.float32)
                emb = np.zeros((B, 144), np.float32)
                ok(lib.afx_conformer_forward(h, ptr(tok), B, T, ptr(logits), ptr(emb), ptr(ws_for(nh)), nh, None), "conformer forward")
                for key in (b"fuse_conformer", b"conf_attn_mfma", b"posconv_sliding", b"fuse_conv_ln"):
                    ok(lib.afx_engine_set(h, key, 0), "engine_set")
                ok(lib.afx_forward(h, ptr(wave), B, Ls, ptr(logits), ptr(ws), n, None), "per-op forward")
                fails(lib.afx_engine_set(h, b"bogus", 0), "unknown key")
        else:
            feats = np.zeros((B, T, 1024), np.float32)
            n = lib.afx_workspace_bytes(h, B, Ls)
            ok(lib.afx_ssl_forward(h, ptr(wave), B, Ls, ptr(feats), ptr(ws_for(n)), n, None), "ssl forward")
            fails(lib.afx_forward(h, ptr(wave), B, Ls, ptr(logits), ptr(ws_for(n)), n, None), "SSL feature extractor")
            lens = (I * B)(16000, 9000, 400)
            frames = (I * B)()
            nr = lib.afx_ragged_workspace_bytes(h, B, Ls)
            ok(lib.afx_ssl_forward_ragged(h, ptr(wave), B, Ls, lens, ptr(feats), frames, ptr(ws_for(nr)), nr, None), "ragged ssl")
            assert list(frames) == [49, 27, 1], list(frames)
        lib.afx_destroy(h)

# MyConformer alone (no trunk in the handle)
h = create(HEAD, blocks=1)
load(h, "conformer", skip=("ssl_model.",))
ok(lib.afx_finalize(h, None), "head-only finalize")
T = 49
tok, emb = np.zeros((B, T, 144), np.float32), np.zeros((B, 144), np.float32)
nh = lib.afx_head_workspace_bytes(h, B, T)
ok(lib.afx_conformer_forward(h, ptr(tok), B, T, ptr(logits), ptr(emb), ptr(ws_for(nh)), nh, None), "MyConformer forward")
fails(lib.afx_forward(h, ptr(wave), B, Ls, ptr(logits), ptr(ws_for(nh)), nh, None), "Conformer blocks only")
lib.afx_destroy(h)

# weight-store error paths
h = create(CONF)
a = np.zeros((3, 3), np.float32)
shape = (C.c_int64 * 2)(3, 3)
fails(lib.afx_load_weight(h, b"ssl_model.model.post_extract_proj.weight", ptr(a), shape, 2, None), "elements, expected")
fails(lib.afx_load_weight(h, b"ssl_model.model.feature_extractor.conv_layers.9.0.weight", ptr(a), shape, 2, None), "out of range")
fails(lib.afx_load_weight(h, b"conformer.encoder_blocks.7.ff1.fn.norm.weight", ptr(a), shape, 2, None), "out of range")
ok(lib.afx_load_weight(h, b"module.ssl_model.model.quantizer.vars", ptr(a), shape, 2, None), "ignored key")
ok(lib.afx_load_weight(h, b"some.buffer.num_batches_tracked", ptr(a), shape, 2, None), "ignored key")
# a tensor reloaded with another size releases its first buffer
ok(lib.afx_load_weight(h, b"LL.bias", ptr(a), shape, 2, None), "raw store")
b2 = np.zeros(144, np.float32)
ok(lib.afx_load_weight(h, b"LL.bias", ptr(b2), (C.c_int64 * 1)(144), 1, None), "raw store, new size")
fails(lib.afx_finalize(h, None), "missing weight")
lib.afx_destroy(h)
bad = Config(CONF, 1, 0, 144, 4, 31, 4, 0, 0.97, 0)
hh = P()
assert lib.afx_create(C.cast(C.byref(bad), SIGNATURES["afx_create"][1][0]), C.byref(hh)) != 0 and b"at least 1" in lib.afx_last_error()
bad = Config(CONF, 1, 1, 145, 4, 31, 4, 0, 0.97, 0)
assert lib.afx_create(C.cast(C.byref(bad), SIGNATURES["afx_create"][1][0]), C.byref(hh)) != 0 and b"bad Conformer" in lib.afx_last_error()

# single-kernel entry points: shape contracts and launch-grid arithmetic of the launchers
M, N, K = 300, 512, 1024
A, Wt = np.zeros((M, K), np.float16), np.zeros((N, K), np.float16)
of = np.zeros((M, N), np.float32)
for m_ in (1, 300, 12736):
    Am, om = np.zeros((m_, K), np.float16), np.zeros((m_, N), np.float32)
    for n_ in (144, 512, 1024, 3072):
        Wn, on = np.zeros((n_, K), np.float16), np.zeros((m_, n_), np.float32)
        ok(lib.afx_k_gemm(1, ptr(Am), K, ptr(Wn), K, m_, n_, K, None, 0, 1.0, None, 0, ptr(on), n_, None, 0, None), "k_gemm")
fails(lib.afx_k_gemm(1, ptr(A), K, ptr(Wt), K, M, N, 100, None, 0, 1.0, None, 0, ptr(of), N, None, 0, None), "multiple of 64")
A32, W32 = np.zeros((M, K), np.float32), np.zeros((N, K), np.float32)
ok(lib.afx_k_gemm(3, ptr(A32), K, ptr(W32), K, M, N, K, None, 0, 1.0, None, 0, ptr(of), N, None, 0, None), "k_gemm fp16x3")
fails(lib.afx_k_gemm(1, ptr(A), K, ptr(Wt), K, M, N, K, None, 0, 1.0, None, 0, None, N, None, 0, None), "no output")
assert lib.afx_shim_launch_count() > 500
print("ASAN_DRIVE_OK launches", lib.afx_shim_launch_count())
