"""ctypes binding of libafx.so (the C ABI declared in include/afx.h).

There is deliberately no fallback: if the shared library is missing, or it
reports an error, the call raises.  Build it with ``__graft_entry__.build()``
or ``make -C real-time-deepfake-speech-detection_amd/csrc``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AFX_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libafx.so")

ARCH_SSL, ARCH_XLSR_AASIST, ARCH_CONFORMER, ARCH_CONFORMER_HEAD = 0, 1, 2, 3
DT_BF16, DT_FP16, DT_FP32, DT_FP16X3 = 0, 1, 2, 3
ACT_NONE, ACT_GELU, ACT_SWISH, ACT_SELU = 0, 1, 2, 3


class AfxError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("arch", C.c_int), ("dtype", C.c_int), ("n_layers", C.c_int),
                ("conf_emb", C.c_int), ("conf_heads", C.c_int), ("conf_kernel", C.c_int),
                ("conf_blocks", C.c_int), ("pre_emphasis", C.c_int), ("pre_emphasis_coef", C.c_float),
                ("extractor_mode", C.c_int)]


_P, _I, _L, _F, _Z = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_size_t

# symbol -> (restype, argtypes); must list every function include/afx.h declares
SIGNATURES = {
    "afx_create": (_I, [C.POINTER(Config), C.POINTER(_P)]),
    "afx_destroy": (None, [_P]),
    "afx_last_error": (C.c_char_p, []),
    "afx_version": (C.c_char_p, []),
    "afx_build_id": (C.c_char_p, []),
    "afx_hip_versions": (_I, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "afx_load_weight": (_I, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), _I, _P]),
    "afx_finalize": (_I, [_P, _P]),
    "afx_num_frames": (_I, [_I]),
    "afx_workspace_bytes": (_Z, [_P, _I, _I]),
    "afx_head_workspace_bytes": (_Z, [_P, _I, _I]),
    "afx_forward": (_I, [_P, _P, _I, _I, _P, _P, _Z, _P]),
    "afx_ssl_forward": (_I, [_P, _P, _I, _I, _P, _P, _Z, _P]),
    "afx_trunk_forward": (_I, [_P, _P, _I, _I, _P, _Z, _P]),
    "afx_head_from_workspace": (_I, [_P, _I, _I, _P, _P, _Z, _P]),
    "afx_head_forward": (_I, [_P, _P, _I, _I, _P, _P, _Z, _P]),
    "afx_ragged_workspace_bytes": (_Z, [_P, _I, _I]),
    "afx_forward_ragged": (_I, [_P, _P, _I, _I, C.POINTER(C.c_int), _P, _P, _Z, _P]),
    "afx_ssl_forward_ragged": (_I, [_P, _P, _I, _I, C.POINTER(C.c_int), _P, C.POINTER(C.c_int), _P, _Z, _P]),
    "afx_tail_workspace_bytes": (_Z, [_P, _I, _I]),
    "afx_tail_forward": (_I, [_P, _P, _I, _I, _P, _P, _Z, _P]),
    "afx_tail_forward_strided": (_I, [_P, _P, _L, _I, _I, _P, _P, _Z, _P]),
    "afx_conformer_forward": (_I, [_P, _P, _I, _I, _P, _P, _P, _Z, _P]),
    "afx_engine_set": (_I, [_P, C.c_char_p, _I]),
    "afx_kv_create": (_I, [_P, _I, C.POINTER(_P)]),
    "afx_kv_destroy": (None, [_P]),
    "afx_kv_state_bytes": (_Z, [_P]),
    "afx_kv_workspace_bytes": (_Z, [_P, _I]),
    "afx_kv_step": (_I, [_P, _P, _I, _P, _P, _Z, _P]),
    "afx_check_finite": (_I, [_P, _P]),
    "afx_enable_taps": (_I, [_P, _I]),
    "afx_tap": (_I, [_P, C.c_char_p, _P, _Z, C.POINTER(_Z), _P]),
    "afx_profile_begin": (_I, [_P]),
    "afx_profile_end": (_I, [_P, _I, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    "afx_profile_num_classes": (_I, []),
    "afx_profile_class_name": (C.c_char_p, [_I]),
    "afx_debug_set": (_I, [C.c_char_p, _I]),
    "afx_k_gemm": (_I, [_I, _P, _L, _P, _L, _I, _I, _I, _P, _I, _F, _P, _L, _P, _L, _P, _L, _P]),
    "afx_k_conv_gemm": (_I, [_I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "afx_k_conv_ln_act": (_I, [_I, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _F, _I, _P, _P, _P]),
    "afx_k_pack_linear": (_I, [_I, _P, _I, _I, _I, _P, _P]),
    "afx_k_pack_conv": (_I, [_I, _P, _I, _I, _I, _P, _P]),
    "afx_k_conv0": (_I, [_I, _P, _I, _I, _P, _P, _P, _P, _I, _F, _P, _P]),
    "afx_k_conv0_pack_bytes": (_Z, []),
    "afx_k_conv0_pack": (_I, [_P, _P, _P, _P]),
    "afx_k_conv0_packed": (_I, [_I, _P, _I, _I, _P, _P, _P, _P, _P, _I, _F, _P, _P]),
    "afx_k_pre_emphasis": (_I, [_P, _I, _I, _F, _P, _P]),
    "afx_k_tile_crop": (_I, [_P, _P, _P, _I, _I, _P, _P]),
    "afx_k_rownorm": (_I, [_I, _P, _L, _I, _I, _P, _P, _F, _I, _P, _L, _P, _L, _P]),
    "afx_k_mhsa": (_I, [_I, _P, _P, _I, _I, _I, _P]),
    "afx_k_conf_attn": (_I, [_I, _P, _L, _P, _L, _P, _I, _I, _I, _I, _I, _P, _L, _P]),
    "afx_k_conf_attn_mfma": (_I, [_I, _P, _L, _P, _L, _P, _I, _I, _I, _I, _I, _P, _L, _P]),
    "afx_k_conf_dwconv": (_I, [_I, _P, _L, _P, _P, _P, _P, _I, _I, _I, _I, _P, _L, _P]),
    "afx_aasist_error": (C.c_char_p, []),
    "afx_k_gat": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P]),
    "afx_k_hgat": (_I, [_P, _I, _P, _I, _I, _I, _I, C.POINTER(_P), _F, _P, _L, _P, _P, _P, _P, _P]),
    "afx_k_resblock_scratch_floats": (_Z, [_I, _I, _I, _I, _I]),
    "afx_k_resblock": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "afx_k_graph_pool": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P]),
}

_lib = None


def lib():
    """The loaded library; raises AfxError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AfxError(f"{LIB_PATH} is missing: the HIP library has not been built "
                           "(run __graft_entry__.build()); there is no CPU fallback")
        # torch first: its wheel bundles a HIP / HSA runtime of its own, and the process must end up with ONE.  Loaded after
        # torch, libafx.so's libamdhip64 / libhsa-runtime64 resolve to the copies torch already holds; loaded before it, the
        # system copies come in as a second runtime and whichever initialises second sees no device (found by running
        # __graft_entry__.build() and smoke() in one process on a GPU box).
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _check_runtime(l)
        _lib = l
        # (no environment hook: the A/B knobs are reached through afx_debug_set / Engine.set by the tools and tests that
        # own the process, never by a stray variable in a product run)
    return _lib


def _check_runtime(l):
    """libafx.so is built by the system's hipcc and runs on whatever HIP runtime the process resolved first -- with torch
    loaded, the copy bundled in the torch wheel.  The code objects and launch stubs of one major version do not belong on the
    runtime of another: refuse it.  A runtime OLDER than the toolchain inside one major version (this image: hipcc 7.2 on the
    wheel's 7.0) works as long as the library sticks to what both know -- said once, as a warning, so that a 'no device' or a
    launch failure after an image change has a first suspect.  Build with the ROCm the torch wheel carries where possible."""
    b, r = C.c_int(0), C.c_int(0)
    if l.afx_hip_versions(C.byref(b), C.byref(r)) != 0:
        raise AfxError(l.afx_last_error().decode())
    bm, rm = b.value // 10_000_000, r.value // 10_000_000
    if bm != rm:
        raise AfxError(f"libafx.so was built with HIP {bm}.x (HIP_VERSION {b.value}) but this process runs HIP runtime {rm}.x "
                       f"({r.value}): rebuild the library with the ROCm of the torch wheel (make -C csrc HIPCC=...)")
    if r.value < b.value:
        import warnings
        warnings.warn(f"libafx.so: built with HIP {b.value}, running on the older runtime {r.value} (same major version) -- "
                      "fine on this image; after an image change rebuild with the runtime's own toolchain first", RuntimeWarning, stacklevel=3)


def check(rc):
    if rc != 0:
        raise AfxError(lib().afx_last_error().decode())


def ptr(t):
    """Device/host address of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    """hipStream_t of torch's current stream on ``device`` (default: the current device)."""
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def call_on(t, fn, *args):
    """``fn(*args, stream)`` with the device ``t`` lives on made current and ``stream`` = torch's current stream ON THAT
    DEVICE: a single-kernel entry point launches where its operands are, whatever torch's current device is (the
    reference passes ``device=rank`` and never calls ``torch.cuda.set_device``, main.py:48)."""
    import torch
    with torch.cuda.device(t.device):
        return fn(*args, stream_ptr(t.device))

