"""The reference's own unified-API tests (gtest/api/api_gtest.cpp:380-932: API_setup, API_compress, API_decompress) restated
case by case against the GPU library, for the codecs it backs.  Same inputs (800 random bytes into a 1600-byte output, the
reference's TestLoad(800, 1600, true)), same expectations; where this library deviates (Zstd levels above 4 have no GPU
parse) the test says so instead of hiding the case."""
import ctypes as C
import random

import pytest

import aocl_compression_amd as a
from aocl_compression_amd import api

pytestmark = pytest.mark.gpu
LZ4, LZ4HC, LZMA, BZIP2, SNAPPY, ZLIB, ZSTD = range(7)
# (lower, upper, default) per method, api_gtest.cpp:74-83; the GPU-backed subset of Zstd's 1..22
LEVELS = {LZ4: (0, 0, 0), SNAPPY: (0, 0, 0), ZSTD: (1, 22, 9)}
ZSTD_GPU_LEVELS = (1, 2, 3)      # for the 800-byte input of these tests: level 4 is "greedy" at <= 16 KiB (clevels.h:114), no GPU parse
ALGOS = [LZ4, SNAPPY, ZSTD]


@pytest.fixture(scope="module")
def lib(eng):
    return eng.lib


def _load(n=800, out=1600, seed=1):
    rnd = random.Random(seed)
    inp = C.create_string_buffer(bytes(rnd.randrange(255) for _ in range(n)), n)      # rand() % 255 (api_gtest.cpp:211-214)
    return inp, C.create_string_buffer(out)


def _desc(level, **kw):
    d = api.LlcDesc()                                                                  # reset_ACD (api_gtest.cpp:300-321)
    d.level = level; d.optVar = 0; d.optOff = 0; d.measureStats = 0
    for k, v in kw.items():
        setattr(d, k, v)
    return d


def _default_level(algo):
    return 1 if algo == ZSTD else LEVELS[algo][2]      # the reference's default for Zstd (9, "lazy2") has no GPU backend


def _run(lib, algo, d, stats=False):
    """run_test (api_gtest.cpp:550-564): setup, compress, decompress, validate, destroy"""
    inp, out = _load()
    d.inBuf = C.cast(inp, C.c_void_p); d.inSize = 800; d.outBuf = C.cast(out, C.c_void_p); d.outSize = 1600
    assert lib.aocl_llc_setup(C.byref(d), algo) == 0
    c = lib.aocl_llc_compress(C.byref(d), algo)
    assert c > 0
    if stats:                                                                          # compress_passed_ms (:520-528)
        assert d.cSize == c and d.cTime > 0 and d.cSpeed > 0
    back = C.create_string_buffer(800)
    d.inBuf = C.cast(out, C.c_void_p); d.inSize = c; d.outBuf = C.cast(back, C.c_void_p); d.outSize = 800
    r = lib.aocl_llc_decompress(C.byref(d), algo)
    assert r == 800 and back.raw == inp.raw                                            # decompress_and_validate (:530-548)
    if stats:
        assert d.dSize == 800 and d.dTime > 0 and d.dSpeed > 0
    lib.aocl_llc_destroy(C.byref(d), algo)
    return c


# ---------------------------------------------------------------- API_setup (:380-470)
@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("opt_off", [0, 1])
def test_setup(lib, algo, opt_off):
    d = _desc(_default_level(algo), optOff=opt_off)
    assert lib.aocl_llc_setup(C.byref(d), algo) == 0
    assert (d.workBuf is not None) == (algo == ZSTD)               # "zstd saves params and context" (:771-778)
    lib.aocl_llc_destroy(C.byref(d), algo)
    inp, out = _load()
    d = _desc(_default_level(algo), optOff=opt_off)
    d.inBuf = C.cast(inp, C.c_void_p); d.inSize = 800; d.outBuf = C.cast(out, C.c_void_p); d.outSize = 1600   # ioSet_common_3
    assert lib.aocl_llc_setup(C.byref(d), algo) == 0
    lib.aocl_llc_destroy(C.byref(d), algo)


def test_setup_invalid_and_excluded(lib):
    d = _desc(0)
    assert lib.aocl_llc_setup(C.byref(d), 7) == -4                  # AOCL_COMPRESSOR_ALGOS_NUM -> ERR_UNSUPPORTED_METHOD (:444-451)
    for algo in (LZMA, BZIP2, ZLIB):
        assert lib.aocl_llc_setup(C.byref(_desc(1)), algo) == -3    # ERR_EXCLUDED_METHOD, as a build with AOCL_EXCLUDE_<codec> (:453-485)


# ---------------------------------------------------------------- API_compress (:570-735)
@pytest.mark.parametrize("algo", ALGOS)
def test_compress_default_opt_on_off_stats_memlimit(lib, algo):
    lvl = _default_level(algo)
    _run(lib, algo, _desc(lvl))                                      # defaultOptOn_common_1
    _run(lib, algo, _desc(lvl, optOff=1))                            # defaultOptOff_common_2
    _run(lib, algo, _desc(lvl, measureStats=1), stats=True)          # measureStats_common_3
    for mem in (1 << 9, 1 << 30, 1 << 31):                           # memLimit_common_4: ignored by every method
        _run(lib, algo, _desc(lvl, memLimit=mem))


@pytest.mark.parametrize("algo", ALGOS)
def test_compress_argument_cases(lib, algo):
    lvl = _default_level(algo)
    inp, out = _load()

    def compress(**kw):
        d = _desc(lvl)
        d.inBuf = C.cast(inp, C.c_void_p); d.inSize = 800; d.outBuf = C.cast(out, C.c_void_p); d.outSize = 1600
        for k, v in kw.items():
            setattr(d, k, v)
        assert lib.aocl_llc_setup(C.byref(d), algo) == 0
        r = lib.aocl_llc_compress(C.byref(d), algo)
        lib.aocl_llc_destroy(C.byref(d), algo)
        return r

    assert compress(inBuf=None) < 0                                  # inpNull_common_5
    r = compress(inSize=0)                                           # inpSzZero_common_6
    assert r == 1 if algo in (LZ4, SNAPPY) else r > 0
    assert compress(outBuf=None) < 0                                 # outNull_common_7
    assert compress(outSize=0) < 0                                   # outSzZero_common_8


def test_compress_all_levels(lib):
    """levelsOptOn / levelsOptOff (:706-735).  The reference runs Zstd 1..22; here 1..3 succeed on this 800-byte input (level 4 too above 16 KiB;
    identical bytes to the reference are checked in tests/test_zstd_api.py) and the others fail with ERR_COMPRESSION_FAILED - the documented deviation."""
    for opt_off in (0, 1):
        for algo in (LZ4, SNAPPY):
            _run(lib, algo, _desc(0, optOff=opt_off))
        for level in range(1, 23):
            if level in ZSTD_GPU_LEVELS:
                _run(lib, ZSTD, _desc(level, optOff=opt_off))
            else:
                inp, out = _load()
                d = _desc(level, optOff=opt_off)
                d.inBuf = C.cast(inp, C.c_void_p); d.inSize = 800; d.outBuf = C.cast(out, C.c_void_p); d.outSize = 1600
                assert lib.aocl_llc_setup(C.byref(d), ZSTD) == 0
                assert lib.aocl_llc_compress(C.byref(d), ZSTD) == -2
                lib.aocl_llc_destroy(C.byref(d), ZSTD)


# ---------------------------------------------------------------- API_decompress (:742-932)
@pytest.mark.parametrize("algo", ALGOS)
def test_decompress_cases(lib, algo):
    lvl = _default_level(algo)
    inp, out = _load()
    d = _desc(lvl)
    d.inBuf = C.cast(inp, C.c_void_p); d.inSize = 800; d.outBuf = C.cast(out, C.c_void_p); d.outSize = 1600
    assert lib.aocl_llc_setup(C.byref(d), algo) == 0
    c = lib.aocl_llc_compress(C.byref(d), algo)
    assert c > 0
    back = C.create_string_buffer(800)

    def decompress(**kw):
        d.inBuf = C.cast(out, C.c_void_p); d.inSize = c; d.outBuf = C.cast(back, C.c_void_p); d.outSize = 800
        for k, v in kw.items():
            setattr(d, k, v)
        return lib.aocl_llc_decompress(C.byref(d), algo)

    assert decompress() == 800 and back.raw == inp.raw               # defaultOptOn_common_1
    assert decompress(inBuf=None) < 0                                # inpNull_common_1
    assert decompress(inSize=0) <= 0                                 # inpSzZero_common_1
    assert decompress(outBuf=None) < 0                               # outNull_common_1
    assert decompress(outSize=0) < 0                                 # outSzZero_common_1
    rnd = random.Random(7)                                           # invalidCprData_common_1: 16 bytes in the middle
    bad = bytearray(out.raw[:c])
    for i in range(c // 2, c // 2 + 16):
        bad[i] = rnd.randrange(255)
    badbuf = C.create_string_buffer(bytes(bad), c)
    r = decompress(inBuf=C.cast(badbuf, C.c_void_p))
    assert r < 0 or back.raw[:r] != inp.raw[:r]
    lib.aocl_llc_destroy(C.byref(d), algo)


# ---------------------------------------------------------------- API_compress_MT (gtest/threads/threads_gtest.cpp:281-433, 520-541)
@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("threads", [(7, 6), (6, 7), (1000, 1001)])
def test_compress_mt_any_thread_counts(lib, algo, threads):
    """the reference's threaded build: compress with one thread count, decompress with another (also beyond the machine's) -
    here numThreads > 1 asks for the RAP path, the partitions are GPU chunks whatever the count; the stream carries the RAP
    frame header the reference validates (:527-534) and decodes under any thread count"""
    import struct
    lib.aocl_get_rap_frame_bound_mt.restype = C.c_int32
    n = (1 << 20) + 12345
    raw = a.corpus.make("text", n // 2, seed=71).tobytes() + bytes(random.Random(5).randrange(255) for _ in range(n - n // 2))
    inp = C.create_string_buffer(raw, n)
    cap = 2 * n + lib.aocl_get_rap_frame_bound_mt()                    # callers add the bound (test/codec_bench.c:499-506)
    out = C.create_string_buffer(cap)
    d = _desc(_default_level(algo), numThreads=threads[0])
    d.inBuf = C.cast(inp, C.c_void_p); d.inSize = n; d.outBuf = C.cast(out, C.c_void_p); d.outSize = cap
    assert lib.aocl_llc_setup(C.byref(d), algo) == 0
    c = lib.aocl_llc_compress(C.byref(d), algo)
    assert c > 0
    head = out.raw[8:24] if algo == ZSTD else out.raw[:16]            # Zstd: the frame rides in a skippable frame
    magic, flen, k, child = struct.unpack("<qIHH", head)
    assert magic == 0x434C4C5F4C434F41 and flen == 16 + 12 * k and k > 1 and child == 0
    back = C.create_string_buffer(n)
    d.numThreads = threads[1]
    d.inBuf = C.cast(out, C.c_void_p); d.inSize = c; d.outBuf = C.cast(back, C.c_void_p); d.outSize = n
    assert lib.aocl_llc_decompress(C.byref(d), algo) == n and back.raw == raw
    lib.aocl_llc_destroy(C.byref(d), algo)
