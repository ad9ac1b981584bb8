"""The C-ABI library loads without a GPU and exports every symbol the headers in include/ declare; host-side
argument checks and pure arithmetic work on CPU; compute entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import aocl_compression_amd as a
from aocl_compression_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    for h in ("aocl_gpu_codec.h", "aocl_compression.h"):
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names |= set(re.findall(r"\b(aoclgpu_[a-z0-9_]+|aocl_llc_[a-z_]+)\s*\(", txt))
    return names


def test_library_exports_every_declared_symbol():
    lib = api.load_library()
    declared = _declared()
    assert len(declared) >= 20
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    assert set(api.EXPORTS) <= declared | {"aoclgpu_time_batch"}


def test_versions_and_bounds():
    lib = api.load_library()
    assert lib.aocl_llc_version().startswith(b"AOCL-Compression 4.2.0")
    for n in (0, 1, 1000, 65536, 131072, 262144, 1 << 20):
        for codec in (a.LZ4, a.SNAPPY, a.ZSTD):
            assert lib.aoclgpu_compress_bound(codec, n) == a.compress_bound(codec, n)
    assert a.compress_bound(a.LZ4, 262144) == 263188 and a.compress_bound(a.SNAPPY, 65536) == 76490


def test_struct_layouts_match_the_reference_abi():
    # aocl_compression_desc on LP64 (api/aocl_compression.h:125-152): 3 pointers, 4 size_t, 2 int, size_t, int, 4 u64, 2 float, 2 int
    assert C.sizeof(api.LlcDesc) == 128
    assert api.LlcDesc.inSize.offset == 24 and api.LlcDesc.measureStats.offset == 72 and api.LlcDesc.cSize.offset == 80
    assert api.LlcDesc.optOff.offset == 120
    assert C.sizeof(api.Batch) == 96


def test_argument_checks_without_gpu():
    lib = api.load_library()
    assert lib.aoclgpu_lz4_compress_chunks(None, None) == -1            # AOCLGPU_ERR_ARG
    b = api.Batch(nchunks=3, total_bytes=10, chunk_bytes=4)            # NULL buffers
    assert lib.aoclgpu_snappy_compress_chunks(C.byref(b), None) == -1
    assert b"NULL" in lib.aoclgpu_last_error()
    d = api.LlcDesc()
    assert lib.aocl_llc_setup(C.byref(d), 99) == -4                    # ERR_UNSUPPORTED_METHOD (api/api.cpp:133-138)
    assert lib.aocl_llc_setup(C.byref(d), 2) == -3                     # LZMA: ERR_EXCLUDED_METHOD (no backend)


def test_frame_entry_points_check_their_arguments_without_gpu():
    """aoclgpu_zstd_compress_frames validates the parameters before it touches the device; the mode switches are plain host state"""
    lib = api.load_library()
    one = (C.c_uint64 * 2)(0, 16)
    out = (C.c_int64 * 1)(0)
    buf = C.create_string_buffer(64)

    def frames(**kw):
        f = api.ZstdFrames(C.addressof(buf), C.addressof(buf), C.addressof(one), C.addressof(one), C.addressof(out), 1, 19, 14, 6, 0, 1, 0, 0)
        for k, v in kw.items():
            setattr(f, k, v)
        return lib.aoclgpu_zstd_compress_frames(C.byref(f), None)

    for bad in (dict(windowLog=9), dict(windowLog=32), dict(hashLog=5), dict(hashLog=25), dict(minMatch=3), dict(minMatch=8), dict(targetLength=(1 << 17) + 1),
                dict(strategy=0), dict(strategy=3), dict(strategy=2, chainLog=5), dict(strategy=2, chainLog=25), dict(src_off=None), dict(out_len=None)):
        assert frames(**bad) == -1, bad                                    # AOCLGPU_ERR_ARG
        assert b"zstd frames" in lib.aoclgpu_last_error()
    assert frames(nframes=0) == 0                                          # nothing to do: no device needed
    assert lib.aoclgpu_zstd_compress_frames(None, None) == -1
    lib.aoclgpu_zstd_large_input_mode.restype = C.c_int
    lib.aoclgpu_lz4f_block_linking.restype = C.c_int
    assert lib.aoclgpu_zstd_large_input_mode() == 0 and lib.aoclgpu_lz4f_block_linking() == 1     # the reference's bytes by default
    lib.aoclgpu_zstd_set_large_input_mode(1); lib.aoclgpu_lz4f_set_block_linking(0)
    assert lib.aoclgpu_zstd_large_input_mode() == 1 and lib.aoclgpu_lz4f_block_linking() == 0
    lib.aoclgpu_zstd_set_large_input_mode(0); lib.aoclgpu_lz4f_set_block_linking(1)


def test_log_lines_follow_aocl_enable_log():
    """AOCL_ENABLE_LOG = ERR | INFO | DEBUG | TRACE (utils/utils.h:92-208): one line per statement, "[LEVEL] : file : function :
    line : text", errors on stderr, the rest on stdout, the level read from the environment at every statement; silent without it"""
    import subprocess, sys
    prog = ("import ctypes as C, sys; sys.path.insert(0, %r); from aocl_compression_amd import api; lib = api.load_library(); "
            "d = api.LlcDesc(); print('rc', lib.aocl_llc_setup(C.byref(d), 9), flush=True); "
            "print('rc', lib.aocl_llc_setup(C.byref(d), 2), flush=True)") % ROOT            # 9: no such method, 2 = LZMA: no backend
    def run(level):
        env = dict(os.environ); env.pop("AOCL_ENABLE_LOG", None)
        if level: env["AOCL_ENABLE_LOG"] = level
        return subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, env=env, timeout=120)
    quiet = run(None)
    assert quiet.returncode == 0 and "[" not in quiet.stdout and "[ERR]" not in quiet.stderr
    err = run("ERR")
    assert "[ERR] : " in err.stderr and "setup failed !! compression method is not supported." in err.stderr
    assert "compression method is excluded from this library build." in err.stderr and "[TRACE]" not in err.stdout and "[INFO]" not in err.stdout
    tr = run("TRACE")
    lines = [l for l in tr.stdout.splitlines() if l.startswith("[")]
    assert any(l.startswith("[TRACE] : ") and l.endswith(" : Enter") and " : aocl_llc_setup : " in l for l in lines)
    assert any(l.startswith("[INFO] : ") and l.endswith("Calling setup method for: lzma") for l in lines)
    assert any(l.endswith(" : Exit") for l in lines) and "[ERR] : " in tr.stderr


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        return
    import pytest
    with pytest.raises(api.EngineError):
        api.Engine(0)
    lib = api.load_library()
    d = api.LlcDesc()
    buf = C.create_string_buffer(b"hello", 5)
    out = C.create_string_buffer(64)
    d.inBuf = C.cast(buf, C.c_void_p); d.inSize = 5; d.outBuf = C.cast(out, C.c_void_p); d.outSize = 64
    assert lib.aocl_llc_compress(C.byref(d), 0) == -2                  # ERR_COMPRESSION_FAILED, never a silent CPU path


def test_core_library_exports_only_the_engine_names():
    """libaocl_compression_gpu_core.so = the same code with the drop-in native names (LZ4_*, ZSTD_*, snappy_*, LZ4F_*) hidden: a
    program that also links upstream libzstd / liblz4 / libsnappy (users of the sequence producer) links this one"""
    import subprocess
    core = os.path.join(ROOT, "aocl-compression_amd", "libaocl_compression_gpu_core.so")
    out = subprocess.run(["nm", "-D", "--defined-only", core], capture_output=True, text=True, check=True).stdout
    names = {l.split()[-1] for l in out.splitlines() if l.split()}
    assert names and all(n.startswith(("aoclgpu_", "aocl_llc_", "aocl_get_rap", "aocl_skip_rap")) for n in names), sorted(names)[:10]
    lib = C.CDLL(core)
    for n in sorted(_declared()) + ["aoclgpu_zstd_sequence_producer"]:
        assert hasattr(lib, n), n
