"""Host-side sanitizer run (the reference has an ASAN build option, CMakeLists.txt:487-493): the library's
host code - the .cpp shims (aocl_api, native_api, zstd_api, lz4frame_api) and the host halves of the .hip files - compiled
with AddressSanitizer + UndefinedBehaviorSanitizer (aocl-compression_amd/build.py --sanitize), loaded into a child pytest
that runs the CPU parts of the ABI / Zstd API / LZ4 frame tests with the ASan runtime preloaded.  No GPU involved."""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_mod():
    spec = importlib.util.spec_from_file_location("_aocl_build", os.path.join(ROOT, "aocl-compression_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_host_code_under_asan_and_ubsan():
    b = _build_mod()
    if not os.path.exists(b.HIPCC):
        pytest.skip("hipcc not available")
    b.build()
    lib = b.build_sanitized()
    rt = b.asan_runtime()
    assert os.path.exists(rt), rt
    env = dict(os.environ)
    env.update({"AOCLGPU_LIB": lib, "LD_PRELOAD": rt, "HIP_VISIBLE_DEVICES": "",
                "ASAN_OPTIONS": "detect_leaks=0:verify_asan_link_order=0:abort_on_error=0:halt_on_error=1",
                "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1"})
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_abi.py"), os.path.join(ROOT, "tests", "test_zstd_api.py"),
                        os.path.join(ROOT, "tests", "test_lz4frame.py"), os.path.join(ROOT, "tests", "test_native_api.py")],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    out = p.stdout + p.stderr
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-3000:]
    assert p.returncode == 0, "child pytest failed:\n" + out[-3000:].replace("...", ". . .")
    assert " passed" in out
