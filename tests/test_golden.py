"""Golden vectors captured from the real reference (tests/golden/make_golden.py): the C oracle must reproduce
them on CPU, the HIP path must reproduce them on the GPU.  No access to /root/reference at run time."""
import hashlib
import json
import os

import pytest

import aocl_compression_amd as a
import oracle_lib as ol

with open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")) as f:
    GOLD = json.load(f)
CODECS = {"lz4": 0, "snappy": 1, "zstd": 2}


def _data(v):
    if v["kind"] == "mix":
        return a.corpus.silesia_mix(v["size"], seed=v["seed"]).tobytes()
    return a.corpus.make(v["kind"], v["size"], seed=v["seed"]).tobytes()


@pytest.mark.parametrize("idx", range(len(GOLD["vectors"])))
def test_oracle_matches_golden(oracle, idx):
    v = GOLD["vectors"][idx]
    raw = _data(v)
    assert hashlib.sha256(raw).hexdigest() == v["input_sha256"], "corpus generator drifted"
    for name, exp in v["out"].items():
        c = oracle.compress(CODECS[name], raw)
        assert len(c) == exp["size"] and hashlib.sha256(c).hexdigest() == exp["sha256"], name
        assert oracle.decompress(CODECS[name], c, len(raw)) == raw


@pytest.mark.parametrize("idx", [i for i, v in enumerate(GOLD["vectors"]) if "zstd_level2" in v])
def test_oracle_matches_golden_zstd_level2(oracle, idx):
    v = GOLD["vectors"][idx]
    raw = _data(v)
    c = oracle.compress(2, raw, level=2)
    assert len(c) == v["zstd_level2"]["size"] and hashlib.sha256(c).hexdigest() == v["zstd_level2"]["sha256"]
    assert oracle.decompress(2, c, len(raw)) == raw


def test_oracle_tiny(oracle):
    for t in GOLD["tiny"]:
        raw = bytes.fromhex(t["input_hex"])
        for name, hx in t["out"].items():
            assert oracle.compress(CODECS[name], raw).hex() == hx, (name, raw)
            assert oracle.decompress(CODECS[name], bytes.fromhex(hx), max(len(raw), 1)) == raw


@pytest.mark.gpu
@pytest.mark.parametrize("idx", range(len(GOLD["vectors"])))
def test_gpu_matches_golden(eng, idx):
    v = GOLD["vectors"][idx]
    raw = _data(v)
    for name, exp in v["out"].items():
        c = eng.compress_host(CODECS[name], raw)
        assert len(c) == exp["size"] and hashlib.sha256(c).hexdigest() == exp["sha256"], name
        assert eng.decompress_host(CODECS[name], c, len(raw)) == raw


@pytest.mark.gpu
@pytest.mark.parametrize("idx", [i for i, v in enumerate(GOLD["vectors"]) if "zstd_level2" in v])
def test_gpu_matches_golden_zstd_level2(eng, idx):
    v = GOLD["vectors"][idx]
    raw = _data(v)
    c = eng.compress_host(2, raw, flags=a.api.FLAG_ZSTD_LEVEL2)
    assert len(c) == v["zstd_level2"]["size"] and hashlib.sha256(c).hexdigest() == v["zstd_level2"]["sha256"]
    assert eng.decompress_host(2, c, len(raw)) == raw


@pytest.mark.gpu
def test_gpu_tiny(eng):
    for t in GOLD["tiny"]:
        raw = bytes.fromhex(t["input_hex"])
        for name, hx in t["out"].items():
            assert eng.compress_host(CODECS[name], raw).hex() == hx, (name, raw)


# ---- round-2 vectors (tests/golden/make_golden_r02.py): Zstd levels 3-4 (double-fast) and LZ4_compress_fast accelerations ----
with open(os.path.join(os.path.dirname(__file__), "golden", "golden_r02.json")) as f:
    GOLD2 = json.load(f)


def _oracle_zstd_params(oracle, raw, e):
    import ctypes as C
    oracle.L.orc_zstd_compress_cparams.restype = C.c_size_t
    oracle.L.orc_zstd_compress_cparams.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int, C.c_uint, C.c_uint, C.c_uint]
    cap = max(ol.bound(ol.ZSTD, len(raw)), 16)
    out = C.create_string_buffer(cap)
    r = oracle.L.orc_zstd_compress_cparams(raw, len(raw), out, cap, e["strategy"], e["hashLog"], e["chainLog"], e["minMatch"])
    return out.raw[:r]


@pytest.mark.parametrize("idx", range(len(GOLD2["vectors"])))
def test_oracle_matches_golden_r02(oracle, idx):
    import ctypes as C
    v = GOLD2["vectors"][idx]
    raw = _data(v)
    assert hashlib.sha256(raw).hexdigest() == v["input_sha256"], "corpus generator drifted"
    for level, e in v["zstd"].items():
        if e["strategy"] != 2:
            continue                                           # level 4 is "greedy" below 16 KiB: outside the path
        c = _oracle_zstd_params(oracle, raw, e)
        assert len(c) == e["size"] and hashlib.sha256(c).hexdigest() == e["sha256"], ("zstd", level)
        assert oracle.decompress(2, c, len(raw)) == raw
    oracle.L.orc_lz4_compress_fast.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int]
    for acc, e in v["lz4_fast"].items():
        cap = ol.bound(ol.LZ4, len(raw))
        out = C.create_string_buffer(cap)
        r = oracle.L.orc_lz4_compress_fast(raw, len(raw), out, cap, int(acc))
        assert r == e["size"] and hashlib.sha256(out.raw[:r]).hexdigest() == e["sha256"], ("lz4", acc)


@pytest.mark.gpu
@pytest.mark.parametrize("idx", range(len(GOLD2["vectors"])))
def test_gpu_matches_golden_r02(eng, idx):
    v = GOLD2["vectors"][idx]
    raw = _data(v)
    for level, e in v["zstd"].items():
        if e["strategy"] != 2:
            continue
        c = eng.compress_host(2, raw, flags=a.api.zstd_level(int(level)))
        assert c is not None and len(c) == e["size"] and hashlib.sha256(c).hexdigest() == e["sha256"], ("zstd", level)
        assert eng.decompress_host(2, c, len(raw)) == raw
    for acc, e in v["lz4_fast"].items():
        c = eng.compress_host(0, raw, flags=a.api.lz4_acceleration(int(acc)))
        assert len(c) == e["size"] and hashlib.sha256(c).hexdigest() == e["sha256"], ("lz4", acc)


# ---- whole-input frames (tests/golden/make_golden_frames.py): Zstd frames of several blocks, LZ4 frames with linked blocks ----
with open(os.path.join(os.path.dirname(__file__), "golden", "golden_frames.json")) as f:
    GOLDF = json.load(f)


def _frame_input(v):
    import zstd_frame_cases as zfc
    raw = zfc.make(v["kind"], v["size"], v["seed"])
    assert hashlib.sha256(raw).hexdigest() == v["input_sha256"], "input generator drifted"
    return raw


@pytest.mark.parametrize("idx", range(len(GOLDF["vectors"])))
def test_oracle_matches_golden_frames(oracle, idx):
    import ctypes as C
    import struct
    import xxhash
    v = GOLDF["vectors"][idx]
    raw = _frame_input(v)
    n = len(raw)
    f = oracle.L.orc_zstd_compress_frame_cp
    f.restype = C.c_size_t
    f.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int] + [C.c_uint] * 5
    for e in v["zstd"]:
        cap = n + (n >> 8) + 64
        out = C.create_string_buffer(cap)
        r = f(raw, n, out, cap, e["strategy"], e["windowLog"], e["hashLog"], e["chainLog"], max(e["minMatch"], 4), e["targetLength"])
        assert r == e["size"] and hashlib.sha256(out.raw[:r]).hexdigest() == e["sha256"], ("zstd", e)
    g = oracle.L.orc_lz4_compress_linked_blocks
    g.restype = C.c_int
    g.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32)]
    for e in v["lz4f_linked"]:
        bs = 1 << (8 + 2 * e["blockSizeID"])
        out = C.create_string_buffer(n + 16)
        lens = (C.c_uint32 * (n // bs + 2))()
        nb = g(raw, n, bs, -e["level"] + 1 if e["level"] < 0 else 1, out, n + 16, lens)
        # the frame around the blocks (lz4frame.c:670-699, :1002): magic, FLG (version 01, no checksums), BD, header checksum
        hdr = bytes([0x40 if n > bs else 0x60, e["blockSizeID"] << 4])    # one block => flagged independent (lz4frame.c:394-395)
        frame = struct.pack("<I", 0x184D2204) + hdr + bytes([(xxhash.xxh32(hdr, seed=0).intdigest() >> 8) & 0xFF])
        at = 0
        for k in range(nb):
            sz = lens[k] & 0x7FFFFFFF
            frame += struct.pack("<I", lens[k]) + out.raw[at:at + sz]
            at += sz
        frame += struct.pack("<I", 0)
        assert len(frame) == e["size"] and hashlib.sha256(frame).hexdigest() == e["sha256"], ("lz4f", e)


@pytest.mark.gpu
@pytest.mark.parametrize("idx", range(len(GOLDF["vectors"])))
def test_gpu_matches_golden_frames(eng, idx):
    """through the reference's own entry points on the GPU library: ZSTD_compress_advanced, LZ4F_compressFrame"""
    import ctypes as C
    import lz4f_lib as lf
    import zstd_frame_cases as zfc
    v = GOLDF["vectors"][idx]
    raw = _frame_input(v)
    n = len(raw)
    L = eng.lib
    L.ZSTD_createCCtx.restype = C.c_void_p
    L.ZSTD_freeCCtx.argtypes = [C.c_void_p]
    L.ZSTD_compress_advanced.restype = C.c_size_t
    L.ZSTD_compress_advanced.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, zfc.ZParams]
    cctx = L.ZSTD_createCCtx()
    try:
        for e in v["zstd"]:
            cp = zfc.CParams(e["windowLog"], e["chainLog"], e["hashLog"], 1, e["minMatch"], e["targetLength"], e["strategy"])
            cap = n + (n >> 8) + 64
            out = C.create_string_buffer(cap)
            r = L.ZSTD_compress_advanced(cctx, out, cap, raw, n, None, 0, zfc.ZParams(cp, zfc.FParams(1, 0, 0)))
            assert r == e["size"] and hashlib.sha256(out.raw[:r]).hexdigest() == e["sha256"], ("zstd", e)
    finally:
        L.ZSTD_freeCCtx(cctx)
    G = lf.bind(L)
    for e in v["lz4f_linked"]:
        c = lf.compress(G, raw, lf.prefs(bsid=e["blockSizeID"], level=e["level"]))
        assert isinstance(c, bytes) and len(c) == e["size"] and hashlib.sha256(c).hexdigest() == e["sha256"], ("lz4f", e)
