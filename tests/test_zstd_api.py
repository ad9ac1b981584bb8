"""The Zstd context family of the reference (zstd.h:261-291,542,581) on the GPU library.
CPU part: ZSTD_getParams / ZSTD_getCParams equal the reference's for every level, size class and both level tables
(pure host arithmetic, checked against oracle/_ref).  GPU part: the call sequence of the reference's own wrapper
(api/codec.cpp:354-436: createCCtx, getParams, CCtx_setParameter, compress_advanced, decompressDCtx) run against this
library gives the reference's bytes for levels 1-4 ("fast" and "double-fast") up to 128 KiB and for every "fast" parameter set at
any size (one frame of blocks), multi-frame streams the reference decodes for double-fast inputs above 128 KiB, and
ZSTD_error_parameter_unsupported for everything without a GPU backend."""
import ctypes as C

import pytest

import aocl_compression_amd as a
from aocl_compression_amd import api, corpus
import oracle_lib as ol


class CParams(C.Structure):
    _fields_ = [(n, C.c_uint) for n in ("windowLog", "chainLog", "hashLog", "searchLog", "minMatch", "targetLength")] + [("strategy", C.c_int)]

    def tup(self):
        return tuple(getattr(self, f[0]) for f in self._fields_)


class FParams(C.Structure):
    _fields_ = [("contentSizeFlag", C.c_int), ("checksumFlag", C.c_int), ("noDictIDFlag", C.c_int)]


class Params(C.Structure):
    _fields_ = [("cParams", CParams), ("fParams", FParams)]


def _bind(L):
    L.ZSTD_getParams.restype = Params
    L.ZSTD_getParams.argtypes = [C.c_int, C.c_ulonglong, C.c_size_t]
    L.ZSTD_getCParams.restype = CParams
    L.ZSTD_getCParams.argtypes = [C.c_int, C.c_ulonglong, C.c_size_t]
    L.ZSTD_createCCtx.restype = C.c_void_p
    L.ZSTD_createDCtx.restype = C.c_void_p
    L.ZSTD_freeCCtx.argtypes = [C.c_void_p]
    L.ZSTD_freeDCtx.argtypes = [C.c_void_p]
    L.ZSTD_CCtx_setParameter.restype = C.c_size_t
    L.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
    for n in ("ZSTD_compress_advanced", "ZSTD_compressCCtx", "ZSTD_decompressDCtx", "ZSTD_compress", "ZSTD_decompress", "ZSTD_compressBound"):
        getattr(L, n).restype = C.c_size_t
    L.ZSTD_compress_advanced.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, Params]
    L.ZSTD_compressCCtx.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    L.ZSTD_decompressDCtx.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    L.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.ZSTD_compressBound.argtypes = [C.c_size_t]
    L.ZSTD_isError.argtypes = [C.c_size_t]
    L.ZSTD_getErrorName.restype = C.c_char_p
    L.ZSTD_getErrorName.argtypes = [C.c_size_t]
    L.aocl_setup_zstd_encode.restype = C.c_void_p
    L.aocl_setup_zstd_encode.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t]
    L.aocl_destroy_zstd_encode.restype = None
    return L


@pytest.fixture(scope="module")
def lib():
    return _bind(api.load_library())


@pytest.fixture(scope="module")
def reflib():
    if not ol.have_ref():
        pytest.skip("reference build (oracle/_ref) not available")
    return _bind(C.CDLL(ol.REF_SO))


SIZES = [0, 1, 63, 64, 100, 513, 4096, 16384, 16385, 65536, 131072, 131073, 262144, 262145, 1 << 20, 100 << 20, (1 << 30) + 1, 1 << 33]


@pytest.mark.parametrize("opt_off", [1, 0])
def test_get_params_equal_the_reference(lib, reflib, opt_off):
    """all levels -7..23 x the size classes, upstream table (optOff = 1) and the AOCL table (optOff = 0, what the reference's
    dispatcher selects once it has run, zstd_compress.c:7390-7425)"""
    for L in (lib, reflib):
        L.aocl_destroy_zstd_encode()
        L.aocl_setup_zstd_encode(opt_off, 3, 0, 1, 0)
    try:
        for level in list(range(-7, 24)) + [-131072, -200000, 100]:
            for n in SIZES:
                got, want = lib.ZSTD_getParams(level, n, 0), reflib.ZSTD_getParams(level, n, 0)
                assert got.cParams.tup() == want.cParams.tup(), (level, n, got.cParams.tup(), want.cParams.tup())
                assert (got.fParams.contentSizeFlag, got.fParams.checksumFlag, got.fParams.noDictIDFlag) == (1, 0, 0)
                assert lib.ZSTD_getCParams(level, n, 0).tup() == reflib.ZSTD_getCParams(level, n, 0).tup()
            for n, d in ((0, 1000), (5000, 100000), (1 << 20, 1 << 16)):
                assert lib.ZSTD_getCParams(level, n, d).tup() == reflib.ZSTD_getCParams(level, n, d).tup(), (level, n, d)
    finally:
        for L in (lib, reflib):
            L.aocl_destroy_zstd_encode()
    assert lib.ZSTD_minCLevel() == reflib.ZSTD_minCLevel() and lib.ZSTD_maxCLevel() == reflib.ZSTD_maxCLevel()
    assert lib.ZSTD_defaultCLevel() == reflib.ZSTD_defaultCLevel()


def test_context_calls_check_their_arguments(lib):
    c = lib.ZSTD_createCCtx(); d = lib.ZSTD_createDCtx()
    assert c and d
    out = C.create_string_buffer(64)
    p = lib.ZSTD_getParams(1, 5, 0)
    assert lib.ZSTD_isError(lib.ZSTD_compress_advanced(None, out, 64, b"hello", 5, None, 0, p))       # cctx == NULL (zstd_compress.c:5396)
    assert lib.ZSTD_getErrorName(lib.ZSTD_compress_advanced(c, None, 64, b"hello", 5, None, 0, p)) == b"Operation on NULL destination buffer"
    assert lib.ZSTD_getErrorName(lib.ZSTD_compress_advanced(c, out, 64, b"hello", 5, b"dict", 4, p)) == b"Unsupported parameter"
    p9 = lib.ZSTD_getParams(9, 5, 0)                                     # lazy2: no GPU backend
    assert lib.ZSTD_getErrorName(lib.ZSTD_compress_advanced(c, out, 64, b"hello", 5, None, 0, p9)) == b"Unsupported parameter"
    assert lib.ZSTD_getErrorName(lib.ZSTD_compressCCtx(c, out, 64, b"hello", 5, 5)) == b"Unsupported parameter"   # greedy
    bad = lib.ZSTD_getParams(1, 5, 0); bad.cParams.windowLog = 40
    assert lib.ZSTD_getErrorName(lib.ZSTD_compress_advanced(c, out, 64, b"hello", 5, None, 0, bad)) == b"Parameter is out of bound"
    nofcs = lib.ZSTD_getParams(1, 5, 0); nofcs.fParams.contentSizeFlag = 0        # frames always carry their content size here
    assert lib.ZSTD_getErrorName(lib.ZSTD_compress_advanced(c, out, 64, b"hello", 5, None, 0, nofcs)) == b"Unsupported parameter"
    step = lib.ZSTD_getParams(1, 5, 0); step.cParams.targetLength = 1 << 18     # beyond ZSTD_TARGETLENGTH_MAX
    assert lib.ZSTD_getErrorName(lib.ZSTD_compress_advanced(c, out, 64, b"hello", 5, None, 0, step)) == b"Parameter is out of bound"
    assert lib.ZSTD_CCtx_setParameter(c, 100, 1) == 0 and lib.ZSTD_CCtx_setParameter(c, 201, 0) == 0
    assert lib.ZSTD_isError(lib.ZSTD_CCtx_setParameter(c, 160, 1))      # ZSTD_c_enableLongDistanceMatching
    assert lib.ZSTD_isError(lib.ZSTD_decompressDCtx(None, out, 64, b"x", 1))
    assert lib.ZSTD_freeCCtx(c) == 0 and lib.ZSTD_freeDCtx(d) == 0 and lib.ZSTD_freeCCtx(None) == 0


def _wrapper_compress(L, data: bytes, level: int, window_log: int = 0):
    """aocl_zstd_compress of the reference (api/codec.cpp:388-422), statement for statement, on library L"""
    cctx = L.ZSTD_createCCtx()
    try:
        zp = L.ZSTD_getParams(level, len(data), 0)
        L.ZSTD_CCtx_setParameter(cctx, 100, level)
        zp.fParams.contentSizeFlag = 1
        if window_log and zp.cParams.windowLog > window_log:
            zp.cParams.windowLog = window_log
            zp.cParams.chainLog = window_log + (1 if zp.cParams.strategy in (6, 7, 8) else 0)
        cap = L.ZSTD_compressBound(len(data))
        out = C.create_string_buffer(max(cap, 1))
        r = L.ZSTD_compress_advanced(cctx, out, cap, data, len(data), None, 0, zp)
        return None if L.ZSTD_isError(r) else out.raw[:r]
    finally:
        L.ZSTD_freeCCtx(cctx)


def _wrapper_decompress(L, blob: bytes, cap: int):
    dctx = L.ZSTD_createDCtx()
    try:
        out = C.create_string_buffer(max(cap, 1))
        r = L.ZSTD_decompressDCtx(dctx, out, cap, blob, len(blob))
        return None if L.ZSTD_isError(r) else out.raw[:r]
    finally:
        L.ZSTD_freeDCtx(dctx)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["text", "binary", "jsonlog"])
def test_reference_wrapper_sequence_on_the_gpu_library(eng, lib, reflib, kind):
    for n in (0, 1, 100, 5000, 16384, 16385, 65536, 100000, 131072):
        data = a.corpus.make(kind, n, seed=21).tobytes() if n else b""
        for level in (1, 2, 3, 4, 0):                                        # 0 = default = 3; 3-4 = double-fast
            want = _wrapper_compress(reflib, data, level)
            got = _wrapper_compress(lib, data, level)
            if level == 4 and 0 < n <= 16384:                                # "greedy" there (clevels.h:114): no GPU parse
                assert got is None
                continue
            assert got is not None and got == want, (kind, n, level)
            assert _wrapper_decompress(lib, want, n) == data
        assert _wrapper_compress(lib, data, 5) is None                        # greedy and up: no backend
    # a window clamp below the input size (aocl_llc's optVar): smaller blocks, a window descriptor in the frame header
    data = a.corpus.make(kind, 65536, seed=22).tobytes()
    for wl in (10, 12, 15, 16):
        assert _wrapper_compress(lib, data, 1, window_log=wl) == _wrapper_compress(reflib, data, 1, window_log=wl), wl
    assert _wrapper_compress(lib, data, 3, window_log=12) == _wrapper_compress(reflib, data, 3, window_log=12)


@pytest.mark.gpu
def test_negative_levels_widen_the_step(eng, lib, reflib):
    """levels below -1: the "fast" parser with stepSize = 1 - level (targetLength = -level, zstd_fast.c:440; literals stay raw,
    zstd_compress_internal.h:609-622) - the reference's bytes at every size, through the frame kernels (both finders: table in
    LDS for a lone frame when hashLog <= 14, table in HBM otherwise), and through ZSTD_compress_advanced with a step of its own"""
    for kind in ("text", "binary", "database", "jsonlog"):
        for n in (0, 5, 1000, 20000, 131072, 131073, 300000, 1500000):
            data = corpus.make(kind, n, seed=11).tobytes() if n else b""
            for level in (-1, -2, -3, -5, -7, -20, -1000, -131072):
                want = _wrapper_compress(reflib, data, level)
                got = _wrapper_compress(lib, data, level)
                assert got is not None and got == want, (kind, n, level)
    data = corpus.make("text", 400000, seed=3).tobytes()
    for tl, hl, mm in ((2, 12, 5), (9, 14, 4), (300, 17, 6), (4, 20, 7)):
        for L in (lib, reflib):
            zp = L.ZSTD_getParams(1, len(data), 0)
            zp.cParams.targetLength = tl; zp.cParams.hashLog = hl; zp.cParams.minMatch = mm
            c = L.ZSTD_createCCtx()
            out = C.create_string_buffer(len(data) + 4096)
            r = L.ZSTD_compress_advanced(c, out, len(out), data, len(data), None, 0, zp)
            L.ZSTD_freeCCtx(c)
            assert not L.ZSTD_isError(r), (tl, hl, mm, L.ZSTD_getErrorName(r))
            if L is lib: got = out.raw[:r]
            else: want = out.raw[:r]
        assert got == want, (tl, hl, mm)


@pytest.mark.gpu
def test_inputs_above_128k(eng, lib, reflib):
    """levels 1-4 at every size: ONE frame of 128 KiB blocks, the reference's bytes ("fast" and "double-fast" rows alike).  The
    caller may ask for 128 KiB pieces as concatenated frames instead (aoclgpu_zstd_set_large_input_mode): a valid stream the
    reference decodes (zstd_decompress.c:1113 reads frame after frame), far faster on large inputs, not the reference's bytes."""
    for n in (131073, 200000, 300000, (1 << 20) + 17, 3 << 20):
        data = a.corpus.make("text", n, seed=23).tobytes()
        for level in (1, 2, 3, 4):
            blob = _wrapper_compress(lib, data, level)
            if reflib.ZSTD_getParams(level, n, 0).cParams.strategy > 2:    # level 4 is "greedy" up to 256 KiB: no GPU parse
                assert blob is None
                continue
            assert blob is not None and blob == _wrapper_compress(reflib, data, level), (n, level)
            assert _wrapper_decompress(lib, blob, n) == data
        out = C.create_string_buffer(16)
        r = lib.ZSTD_compress(out, 16, data, n, 1)
        assert lib.ZSTD_getErrorName(r) == b"Destination buffer is too small"
        cap = lib.ZSTD_compressBound(n)                                    # ZSTD_compress takes the same road
        o1 = C.create_string_buffer(cap)
        r1 = lib.ZSTD_compress(o1, cap, data, n, 3)
        assert o1.raw[:r1] == _wrapper_compress(reflib, data, 3)
    eng.lib.aoclgpu_zstd_set_large_input_mode(1)
    try:
        for level in (1, 3):
            blob = _wrapper_compress(lib, data, level)
            assert blob != _wrapper_compress(reflib, data, level) and blob.startswith(_wrapper_compress(reflib, data[:131072], level))
            assert _wrapper_decompress(reflib, blob, n) == data and _wrapper_decompress(lib, blob, n) == data
    finally:
        eng.lib.aoclgpu_zstd_set_large_input_mode(0)
    assert _wrapper_compress(lib, data, 1) == _wrapper_compress(reflib, data, 1)
    small = C.create_string_buffer(1000)
    ref_blob = _wrapper_compress(reflib, data, 1)
    assert lib.ZSTD_isError(lib.ZSTD_decompress(small, 1000, ref_blob, len(ref_blob)))


@pytest.mark.gpu
def test_advanced_parameters_and_checksums(eng, lib, reflib):
    """ZSTD_compress_advanced with explicit "fast" and "double-fast" parameters that are no level's row, with and without the content checksum
    (ZSTD_writeEpilogue, zstd_compress.c:4700-4706): the reference's bytes at every size"""
    c, rc = lib.ZSTD_createCCtx(), reflib.ZSTD_createCCtx()
    try:
        for i, n in enumerate((0, 5, 1000, 70000, 131072, 131073, 400000, 1500000)):
            data = a.corpus.make("jsonlog", n, seed=60 + i).tobytes() if n else b""
            for (st, wl, hl, mm, tl, ck) in ((1, 20, 12, 5, 0, 0), (1, 11, 9, 4, 1, 1), (1, 17, 18, 7, 0, 1), (1, 25, 21, 3, 0, 0), (1, 14, 15, 6, 0, 1),
                                             (2, 20, 15, 4, 0, 1), (2, 13, 12, 6, 8, 0)):
                cap = lib.ZSTD_compressBound(n) + 8
                outs = []
                for (L, ctx) in ((lib, c), (reflib, rc)):
                    p = L.ZSTD_getParams(1, n, 0)
                    p.cParams.windowLog, p.cParams.hashLog, p.cParams.minMatch, p.cParams.targetLength = wl, hl, mm, tl
                    p.cParams.strategy, p.cParams.chainLog = st, hl - 1
                    p.fParams.checksumFlag = ck
                    out = C.create_string_buffer(cap)
                    r = L.ZSTD_compress_advanced(ctx, out, cap, data, n, None, 0, p)
                    assert not L.ZSTD_isError(r), (n, st, wl, hl, mm, L.ZSTD_getErrorName(r))
                    outs.append(out.raw[:r])
                assert outs[0] == outs[1], (n, st, wl, hl, mm, tl, ck)
                assert _wrapper_decompress(lib, outs[0], n) == data
    finally:
        lib.ZSTD_freeCCtx(c); reflib.ZSTD_freeCCtx(rc)


@pytest.mark.gpu
def test_unified_api_zstd_follows_the_wrapper(eng, lib, reflib):
    """aocl_llc_compress(ZSTD): level and optVar (window clamp) go through the same sequence (csrc/aocl_api.cpp)"""
    data = a.corpus.make("text", 100000, seed=25).tobytes()
    for level, ok in ((1, True), (2, True), (0, True), (3, True), (4, True), (9, False)):
        r, blob, _ = eng.llc_compress(a.ZSTD, data, 200000, level=level)
        if ok:
            assert r > 0 and blob == _wrapper_compress(reflib, data, level)
        else:
            assert r == -2                                                   # ERR_COMPRESSION_FAILED
    big = a.corpus.make("text", 500000, seed=26).tobytes()
    r, blob, _ = eng.llc_compress(a.ZSTD, big, 600000, level=1)
    assert r > 0 and blob == _wrapper_compress(reflib, big, 1)               # one frame of four blocks, the reference's bytes
    r2, back, _ = eng.llc_decompress(a.ZSTD, blob, len(big))
    assert r2 == len(big) and back == big
    assert _wrapper_decompress(reflib, blob, len(big)) == big


@pytest.mark.gpu
def test_content_checksums_are_verified_on_the_host_paths(eng, lib, reflib):
    """frames that carry an XXH64 content checksum (zstd_decompress.c:1087-1094): right checksum accepted, wrong one refused -
    single frame, a frame without content size, and two concatenated frames (decoded in parallel)"""
    def ref_frame(data, level=3, checksum=1, content_size=1):
        cctx = reflib.ZSTD_createCCtx()
        zp = reflib.ZSTD_getParams(level, len(data), 0)
        zp.fParams.checksumFlag = checksum
        zp.fParams.contentSizeFlag = content_size
        cap = reflib.ZSTD_compressBound(len(data))
        out = C.create_string_buffer(cap)
        r = reflib.ZSTD_compress_advanced(cctx, out, cap, data, len(data), None, 0, zp)
        reflib.ZSTD_freeCCtx(cctx)
        assert not reflib.ZSTD_isError(r)
        return out.raw[:r]

    d1 = a.corpus.make("text", 200000, seed=51).tobytes()
    d2 = a.corpus.make("binary", 70000, seed=52).tobytes()
    f1, f2 = ref_frame(d1), ref_frame(d2)
    f3 = ref_frame(d2, content_size=0)
    assert _wrapper_decompress(lib, f1, len(d1)) == d1
    assert _wrapper_decompress(lib, f3, len(d2)) == d2
    assert _wrapper_decompress(lib, f1 + f2, len(d1) + len(d2)) == d1 + d2
    for blob, n in ((f1, len(d1)), (f3, len(d2)), (f1 + f2, len(d1) + len(d2)), (f2 + f1, len(d1) + len(d2))):
        bad = bytearray(blob); bad[-1] ^= 0x40                         # the last frame's stored checksum
        assert _wrapper_decompress(lib, bytes(bad), n) is None
        assert _wrapper_decompress(reflib, bytes(bad), n) is None       # the reference refuses it too
    bad = bytearray(f1 + f2); bad[len(f1) - 2] ^= 1                     # the first frame's checksum
    assert _wrapper_decompress(lib, bytes(bad), len(d1) + len(d2)) is None
    r, back, _ = eng.llc_decompress(a.ZSTD, f1 + f2, len(d1) + len(d2))
    assert r == len(d1) + len(d2) and back == d1 + d2


def test_frame_inspection_equals_the_reference(lib, reflib):
    """ZSTD_findFrameCompressedSize / ZSTD_decompressBound / ZSTD_getDecompressedSize / ZSTD_getFrameContentSize: host
    arithmetic on headers, compared with the reference on its own frames (single- and multi-block, with and without content
    size, concatenated, behind a skippable frame, truncated)"""
    import struct
    for L in (lib, reflib):
        L.ZSTD_findFrameCompressedSize.restype = C.c_size_t
        L.ZSTD_findFrameCompressedSize.argtypes = [C.c_char_p, C.c_size_t]
        for n in ("ZSTD_decompressBound", "ZSTD_getDecompressedSize", "ZSTD_getFrameContentSize"):
            getattr(L, n).restype = C.c_ulonglong
            getattr(L, n).argtypes = [C.c_char_p, C.c_size_t]

    def ref_frame(data, level, content_size):
        cctx = reflib.ZSTD_createCCtx()
        zp = reflib.ZSTD_getParams(level, len(data), 0)
        zp.fParams.contentSizeFlag = content_size
        cap = reflib.ZSTD_compressBound(len(data))
        out = C.create_string_buffer(cap)
        r = reflib.ZSTD_compress_advanced(cctx, out, cap, data, len(data), None, 0, zp)
        reflib.ZSTD_freeCCtx(cctx)
        return out.raw[:r]

    d1 = a.corpus.make("text", 400000, seed=81).tobytes()
    d2 = a.corpus.make("binary", 5000, seed=82).tobytes()
    frames = [ref_frame(d1, 1, 1), ref_frame(d1, 3, 0), ref_frame(d2, 5, 1), ref_frame(b"", 1, 1), ref_frame(d2, 1, 0)]
    skippable = struct.pack("<II", 0x184D2A53, 7) + b"1234567"
    streams = frames + [frames[0] + frames[2], skippable + frames[1] + frames[4], frames[2][:-3], frames[0][:50], b"\x00" * 20, b""]
    for blob in streams:
        for name in ("ZSTD_findFrameCompressedSize", "ZSTD_decompressBound", "ZSTD_getDecompressedSize", "ZSTD_getFrameContentSize"):
            got, want = getattr(lib, name)(blob, len(blob)), getattr(reflib, name)(blob, len(blob))
            if name == "ZSTD_findFrameCompressedSize" and reflib.ZSTD_isError(want):
                assert lib.ZSTD_isError(got), (name, len(blob))                    # both refuse (error codes may differ)
            else:
                assert got == want, (name, len(blob), got, want)


# ---- the dispatcher's state machine, in FRESH processes, against the reference (zstd_compress.c:72, 7429-7463) ----
_SEQ_DRIVER = r'''
import ctypes as C, json, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import test_zstd_api as t
L = t._bind(C.CDLL(sys.argv[2]))
out = []
buf = C.create_string_buffer(4096); src = b"abcdefgh" * 64
def probe():
    p3 = L.ZSTD_getCParams(3, 1 << 20, 0); p4 = L.ZSTD_getCParams(4, 100 << 20, 0)
    out.append([p3.chainLog, p4.windowLog, p4.chainLog])
for op in json.loads(sys.argv[3]):
    if op == "probe": probe()
    elif op == "compress": L.ZSTD_compress(buf, 4096, src, len(src), 1)           # (may fail without a GPU: the dispatcher ran first)
    elif op == "cctx": c = L.ZSTD_createCCtx(); L.ZSTD_compressCCtx(c, buf, 4096, src, len(src), 1); L.ZSTD_freeCCtx(c)
    elif op == "setup0": L.aocl_setup_zstd_encode(0, 0, 0, 0, 0)
    elif op == "setup1": L.aocl_setup_zstd_encode(1, 0, 0, 0, 0)
    elif op == "destroy": L.aocl_destroy_zstd_encode()
print(json.dumps(out))
'''

SEQUENCES = [
    ["probe"],                                                     # a fresh process reads the upstream table
    ["probe", "compress", "probe"],                                # the first compress entry installs the AOCL table
    ["setup1", "probe", "setup0", "probe"],                        # a second setup is ignored while set up
    ["setup1", "destroy", "probe", "setup0", "probe"],             # destroy clears the flag, not the table
    ["compress", "setup1", "probe", "destroy", "setup1", "probe"], # setup after a native compress is ignored until destroy
    ["cctx", "probe", "destroy", "probe"],
]


@pytest.mark.parametrize("seq", SEQUENCES, ids=["-".join(s) for s in SEQUENCES])
@pytest.mark.parametrize("env_off", [False, True], ids=["opt", "AOCL_DISABLE_OPT"])
def test_level_table_state_follows_the_reference(seq, env_off):
    import json, os, subprocess, sys
    if not ol.have_ref():
        pytest.skip("reference build (oracle/_ref) not available")
    env = dict(os.environ)
    env.pop("AOCL_DISABLE_OPT", None)
    if env_off:
        env["AOCL_DISABLE_OPT"] = "ON"
    env["HIP_VISIBLE_DEVICES"] = ""                                 # host logic only: same result with or without a GPU
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for so in (api.lib_path(), ol.REF_SO):
        p = subprocess.run([sys.executable, "-c", _SEQ_DRIVER, root, so, json.dumps(seq)], env=env, capture_output=True, text=True, timeout=120)
        assert p.returncode == 0, p.stderr[-500:]
        res.append(json.loads(p.stdout.strip().splitlines()[-1]))
    assert res[0] == res[1], f"{seq}: library {res[0]} reference {res[1]}"
