"""Zstd frames of any size on the GPU (aoclgpu_zstd_compress_frames) against the oracle's frame (pinned byte for byte against
the reference's ZSTD_compress_advanced in tests/test_oracle_vs_ref.py) and, when the reference build is present, against the
reference itself."""
import ctypes as C

import numpy as np
import pytest

import aocl_compression_amd as a
import oracle_lib as ol
import zstd_frame_cases as zfc

pytestmark = pytest.mark.gpu

PARAMS = ((17, 14, 6), (12, 13, 5), (19, 14, 7), (23, 12, 4), (18, 14, 6), (23, 14, 7), (20, 16, 6))


def oracle_frame(oracle, data, wl, hl, mm, tl=0):
    f = oracle.L.orc_zstd_compress_frame_tl
    f.restype = C.c_size_t
    f.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_uint, C.c_uint, C.c_uint, C.c_uint]
    n = len(data)
    cap = n + (n >> 8) + 64
    out = C.create_string_buffer(cap)
    r = f(data, n, out, cap, wl, hl, mm, tl)
    assert r > 0
    return out.raw[:r]


def oracle_frame_df(oracle, data, wl, hl, cl, mm):
    f = oracle.L.orc_zstd_compress_frame_cp
    f.restype = C.c_size_t
    f.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int] + [C.c_uint] * 5
    n = len(data)
    cap = n + (n >> 8) + 64
    out = C.create_string_buffer(cap)
    r = f(data, n, out, cap, 2, wl, hl, cl, mm, 0)
    assert r > 0
    return out.raw[:r]


def gpu_frames(eng, inputs, wl, hl, mm, caps=None, tl=0, strategy=1, cl=0, flags=0):
    t = eng.torch
    off = np.zeros(len(inputs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in inputs])
    blob = b"".join(inputs)
    src = t.frombuffer(bytearray(blob) if blob else bytearray(1), dtype=t.uint8).to(eng.device)
    dst, doff, lens = eng.zstd_compress_frames(src, off, wl, hl, mm, caps=caps, target_length=tl, strategy=strategy, chain_log=cl, flags=flags)
    host = dst.cpu().numpy().tobytes()
    return [host[doff[i]:doff[i] + lens[i]] for i in range(len(inputs))]


@pytest.mark.parametrize("kind", zfc.KINDS)
def test_frames_bit_exact(eng, oracle, kind):
    """a batch of frames per parameter set: block-boundary sizes, a window below the input, a window below the block size"""
    sizes = zfc.EDGE_SIZES + (3000000, 0, 1, 6, 7, 100, 70000, 131072)
    inputs = [zfc.make(kind, n, seed=200 + i) if n else b"" for i, n in enumerate(sizes)]
    for (wl, hl, mm) in PARAMS:
        # both finders of the fast strategy (table in HBM / table and source ring in LDS, hashLog <= 14), then the launcher's choice
        for flags in ((1, 2, 0) if hl <= 14 else (1,)):
            got = gpu_frames(eng, inputs, wl, hl, mm, flags=flags)
            for data, g in zip(inputs, got):
                want = oracle_frame(oracle, data, wl, hl, mm)
                assert g == want, (kind, len(data), wl, hl, mm, flags, len(g), len(want))
    # the frames decode on the GPU too (multi-block frames, treeless literal blocks, RLE and raw blocks)
    for data, g in zip(inputs[:6], got[:6]):
        out = C.create_string_buffer(len(data))
        r = eng.lib.aoclgpu_zstd_decompress_host_frames(g, len(g), out, len(data))
        assert r == len(data) and out.raw == data, (kind, len(data), r)


DF_PARAMS = ((18, 14, 14, 5), (23, 17, 15, 5), (23, 18, 16, 5), (12, 11, 10, 4), (21, 17, 16, 7), (17, 16, 15, 6))   # (window, long, short, minMatch)


@pytest.mark.parametrize("kind", zfc.KINDS)
def test_double_fast_frames_bit_exact(eng, oracle, kind):
    sizes = zfc.EDGE_SIZES + (3000000, 0, 1, 7, 27, 28, 100, 70000, 131072)
    inputs = [zfc.make(kind, n, seed=600 + i) if n else b"" for i, n in enumerate(sizes)]
    for (wl, hl, cl, mm) in DF_PARAMS:
        got = gpu_frames(eng, inputs, wl, hl, mm, strategy=2, cl=cl)
        for data, g in zip(inputs, got):
            want = oracle_frame_df(oracle, data, wl, hl, cl, mm)
            assert g == want, (kind, len(data), wl, hl, cl, mm, len(g), len(want))


def test_frames_against_reference(eng, ref):
    """the level's own parameters for the size (AOCL table), straight against the reference build"""
    for i, n in enumerate((131073, 262145, 400000, 1 << 20, 2500000)):
        data = zfc.make("mixed", n, seed=300 + i)
        for level in (1, 2, 3, 4):
            cp = zfc.ref_level_cparams(ref, level, n)
            if cp.strategy > 2:
                continue                                   # level 4 is "greedy" up to 256 KiB: no GPU parse
            want = zfc.ref_frame(ref, data, cp)
            got = gpu_frames(eng, [data], cp.windowLog, cp.hashLog, cp.minMatch, strategy=cp.strategy, cl=cp.chainLog)[0]
            assert got == want, (n, level)


def test_target_length_one_leaves_literals_raw(eng, oracle):
    """fast with targetLength 1 (ZSTD level -1, row 0 of the level tables): the same parse, literals never Huffman-coded
    (ZSTD_literalsCompressionIsDisabled, zstd_compress_internal.h:609-622)"""
    inputs = [zfc.make(k, n, seed=400 + i) for i, (k, n) in enumerate((("text", 1000), ("jsonlog", 70000), ("mixed", 400000), ("low", 131073)))]
    for (wl, hl, mm) in ((19, 12, 6), (11, 9, 4), (19, 13, 7)):
        for flags in (1, 2):
            got = gpu_frames(eng, inputs, wl, hl, mm, tl=1, flags=flags)
            for data, g in zip(inputs, got):
                assert g == oracle_frame(oracle, data, wl, hl, mm, tl=1), (len(data), wl, hl, mm, flags)
                assert g != oracle_frame(oracle, data, wl, hl, mm, tl=0) or len(data) < 100


def test_small_capacity_reports_zero(eng, oracle):
    data = zfc.make("text", 300000, seed=9)
    want = oracle_frame(oracle, data, 19, 14, 7)
    got = gpu_frames(eng, [data, data, data], 19, 14, 7, caps=[len(want), len(want) - 1, 10])
    assert got[0] == want and got[1] == b"" and got[2] == b""


def test_bad_parameters_are_refused(eng):
    t = eng.torch
    src = t.zeros(16, dtype=t.uint8, device=eng.device)
    for (wl, hl, mm) in ((9, 14, 6), (32, 14, 6), (19, 25, 6), (19, 14, 3), (19, 14, 8)):
        with pytest.raises(a.EngineError):
            eng.zstd_compress_frames(src, [0, 16], wl, hl, mm)
    with pytest.raises(a.EngineError):
        eng.zstd_compress_frames(src, [0, 16], 19, 14, 6, target_length=(1 << 17) + 1)      # ZSTD_TARGETLENGTH_MAX
    with pytest.raises(a.EngineError):
        eng.zstd_compress_frames(src, [0, 16], 19, 15, 6, flags=2)          # the LDS finder serves hashLog <= 14
    for (strategy, cl) in ((3, 14), (0, 14), (2, 5), (2, 25)):
        with pytest.raises(a.EngineError):
            eng.zstd_compress_frames(src, [0, 16], 19, 14, 6, strategy=strategy, chain_log=cl)
