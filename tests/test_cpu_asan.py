"""The HOST side of libafx under AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are
not available on the pool).  `make asan` compiles the product sources host-only with -fsanitize=address,undefined and
links them against tests/asan/hip_host_shim.cpp (device memory = malloc, kernel launches = no-ops: test infrastructure,
it computes nothing and the product never links it); tests/asan/drive_host.py then walks the weight store, finalize,
workspace carving, every forward entry point (uniform, ragged, head, tail, MyConformer), taps, the profiler, the error
paths and the launchers' shape logic through the C ABI, in a process that preloads the sanitizer runtime."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")


def test_host_side_is_clean_under_asan_and_ubsan(tmp_path):
    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not rt:
        pytest.skip("no AddressSanitizer runtime in this image")
    subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "-j", str(min(8, os.cpu_count() or 2)), "asan"], check=True,
                   capture_output=True)
    lib = os.path.join(PKG, "lib", "libafx_asan.so")
    assert os.path.exists(lib)
    sys.path.insert(0, PKG)
    from afx import synth
    arrays = {}
    for arch, name in (("xlsr_aasist", "XLSR_AASIST"), ("conformer", "ConformerModel")):
        for k, v in synth.model_state_dict(name, n_layers=1, **({"n_encoders": 1} if arch == "conformer" else {})).items():
            if v.dtype.is_floating_point:
                arrays[f"{arch}/{k}"] = v.numpy()
    npz = tmp_path / "weights.npz"
    np.savez(npz, **arrays)
    env = dict(os.environ, LD_PRELOAD=rt[-1], ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asan", "drive_host.py"), lib, str(npz)],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "ASAN_DRIVE_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-6000:])
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-6000:]
