"""SURVEY 8f row 4 (second half): the GPU match finder as a block-level external sequence producer of the REFERENCE's
libzstd (ZSTD_registerSequenceProducer, algos/zstd/lib/zstd.h:4065-4098).  The reference library (oracle/_ref) validates
every sequence (ZSTD_c_validateSequences), runs its own entropy stage, and its decoder must give the input back."""
import ctypes as C
import os

import pytest

import aocl_compression_amd as a
from aocl_compression_amd import api
import oracle_lib as ol

pytestmark = pytest.mark.gpu
ZSTD_c_compressionLevel, ZSTD_c_validateSequences, ZSTD_c_enableSeqProducerFallback, ZSTD_c_searchForExternalRepcodes = 100, 1009, 1014, 1016


@pytest.fixture(scope="module")
def ref():
    if not os.path.exists(ol.REF_SO):
        pytest.skip("reference build not present")
    R = C.CDLL(ol.REF_SO)
    R.ZSTD_createCCtx.restype = C.c_void_p
    R.ZSTD_freeCCtx.argtypes = [C.c_void_p]
    R.ZSTD_CCtx_setParameter.restype = C.c_size_t
    R.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
    R.ZSTD_registerSequenceProducer.restype = None
    R.ZSTD_registerSequenceProducer.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    R.ZSTD_compress2.restype = C.c_size_t
    R.ZSTD_compress2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    R.ZSTD_compress.restype = C.c_size_t
    R.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    R.ZSTD_decompress.restype = C.c_size_t
    R.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    R.ZSTD_isError.argtypes = [C.c_size_t]
    R.ZSTD_compressBound.restype = C.c_size_t
    R.ZSTD_compressBound.argtypes = [C.c_size_t]
    return R


@pytest.mark.parametrize("kind,size", [("text", 1 << 20), ("jsonlog", 300000), ("binary", 131072), ("database", 700001),
                                       ("random", 200000), ("text", 5), ("text", 100)])
def test_reference_zstd_with_gpu_sequences(eng, ref, kind, size):
    # the core library (native names hidden): what a program that links upstream / reference libzstd next to the engine uses
    G = C.CDLL(os.path.join(os.path.dirname(api.lib_path()), "libaocl_compression_gpu_core.so"))
    producer = C.cast(G.aoclgpu_zstd_sequence_producer, C.c_void_p)
    data = a.corpus.make(kind, max(size, 4096)).tobytes()[:size]
    cctx = ref.ZSTD_createCCtx()
    try:
        assert not ref.ZSTD_isError(ref.ZSTD_CCtx_setParameter(cctx, ZSTD_c_compressionLevel, 1))
        assert not ref.ZSTD_isError(ref.ZSTD_CCtx_setParameter(cctx, ZSTD_c_validateSequences, 1))
        assert not ref.ZSTD_isError(ref.ZSTD_CCtx_setParameter(cctx, ZSTD_c_enableSeqProducerFallback, 0))   # a producer error fails the call
        # below level 10 zstd stores external offsets as they come; the fast parse relies on repcodes, so let zstd look for them
        assert not ref.ZSTD_isError(ref.ZSTD_CCtx_setParameter(cctx, ZSTD_c_searchForExternalRepcodes, 1))
        ref.ZSTD_registerSequenceProducer(cctx, None, producer)
        cap = ref.ZSTD_compressBound(size)
        out = C.create_string_buffer(cap)
        r = ref.ZSTD_compress2(cctx, out, cap, data, size)
        assert not ref.ZSTD_isError(r), r
    finally:
        ref.ZSTD_freeCCtx(cctx)
    back = C.create_string_buffer(max(size, 1))
    d = ref.ZSTD_decompress(back, size, out, r)
    assert d == size and back.raw[:size] == data
    # also through this repo's GPU decoder
    assert eng.decompress_host(a.ZSTD, out.raw[:r], size) == data
    if size >= 131072 and kind != "random":
        # the same parse as ZSTD_compress(.., 1) of each 128 KiB block on its own (blocks are parsed without history), with
        # one frame header instead of one per block: never larger than the sum of those frames
        total = 0
        for i in range(0, size, 131072):
            part = data[i:i + 131072]
            plain = C.create_string_buffer(ref.ZSTD_compressBound(len(part)))
            total += ref.ZSTD_compress(plain, len(plain), part, len(part), 1)
        assert r <= total * 1.005, (r, total)
