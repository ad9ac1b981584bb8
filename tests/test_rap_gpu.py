"""RAP streams: the GPU join of the per-chunk outputs into the reference's multi-threaded
container (csrc/rap.hip) against the reference built with AOCL_ENABLE_THREADS (oracle/_ref/libaocl_ref_mt.so).

 * with the reference's partition geometry (T threads -> T equal partitions) the LZ4 and Snappy streams are
   byte-identical to the reference's MT output (lz4.c:2655-2909 last-literals join, snappy.cc:2567-2654 varint join);
 * every stream decodes with a legacy single-threaded decoder after aocl_skip_rap_frame_mt (README.md:341-343,
   zstd_gtest.cpp:589-603) and with the reference's MT decoder;
 * the GPU decoder accepts the reference's MT streams and its own."""
import ctypes as C
import struct

import numpy as np
import pytest
import torch

import aocl_compression_amd as a
import oracle_lib as ol

pytestmark = pytest.mark.gpu
CODECS = {"lz4": (a.LZ4, ol.LZ4), "snappy": (a.SNAPPY, ol.SNAPPY), "zstd": (a.ZSTD, ol.ZSTD)}


def _mt_cap(n, parts):
    return n + n // 6 + 64 + 16 + 12 * parts + 1024


def _entries(stream: bytes, base=0):
    magic, flen, cnt = struct.unpack_from("<qII", stream, base)
    assert magic == 0x434C4C5F4C434F41
    k = cnt & 0xFFFF
    assert flen == 16 + 12 * k
    return [struct.unpack_from("<III", stream, base + 16 + 12 * i) for i in range(k)]


@pytest.mark.parametrize("kind", ["text", "jsonlog", "binary", "database", "random"])
@pytest.mark.parametrize("codec", ["lz4", "snappy"])
def test_stream_is_byte_identical_to_the_reference_mt_build(eng, ref, ref_mt, codec, kind):
    gc, oc = CODECS[codec]
    T, part = 4, 300000
    data = a.corpus.make(kind, T * part, seed=11).tobytes()
    ref_mt.set_threads(T)
    want = ref_mt.compress(oc, data, _mt_cap(len(data), T))
    assert want is not None and len(_entries(want)) == T            # the reference really used T partitions of `part` bytes
    src = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(eng.device)
    got = bytes(eng.rap_compress(gc, src, part).cpu().numpy())
    assert got == want
    # legacy single-threaded decoder after skipping the frame
    skip = ref_mt.skip(got)
    assert skip == 16 + 12 * T
    assert ref.decompress(oc, got[skip:], len(data)) == data
    # GPU decoder on the stream
    dst, r = eng.rap_decompress(gc, torch.frombuffer(bytearray(want), dtype=torch.uint8).to(eng.device), len(data))
    assert r == len(data) and bytes(dst[:r].cpu().numpy()) == data


@pytest.mark.parametrize("kind", ["text", "jsonlog", "binary", "mixed"])
def test_zstd_stream_is_byte_identical_to_the_reference_mt_build(eng, ref, ref_mt, kind):
    """Zstd: the threaded reference cuts the input into num_threads partitions once it spans several windows
    (threads/threads.c:57-97 with chunk = window, zstd_compress.c:5418-5420) and compresses each with the caller's parameters
    as one frame (:5447-5468) - multi-block frames with a window descriptor here (window 128 KiB, partitions of 300 000 bytes)."""
    import zstd_frame_cases as zfc
    T, part = 4, 300000
    data = zfc.make(kind, T * part, seed=12)
    for cp in (zfc.CParams(17, 13, 14, 1, 6, 0, 1), zfc.CParams(17, 15, 16, 1, 5, 0, 2)):
        ref_mt.set_threads(T)
        want = zfc.ref_frame(ref_mt, data, cp, extra=1024)
        ents = _entries(want, 8)
        assert len(ents) == T and all(e[2] == part for e in ents)       # the reference really used T partitions of `part` bytes
        src = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(eng.device)
        got = bytes(eng.zstd_rap_compress(src, [i * part for i in range(T + 1)], cp.windowLog, cp.hashLog, cp.minMatch,
                                          strategy=cp.strategy, chain_log=cp.chainLog).cpu().numpy())
        assert got == want
        assert ref.decompress(ol.ZSTD, got, len(data)) == data
        dst, r = eng.rap_decompress(a.ZSTD, torch.frombuffer(bytearray(got), dtype=torch.uint8).to(eng.device), len(data))
        assert r == len(data) and bytes(dst[:r].cpu().numpy()) == data


@pytest.mark.parametrize("codec", ["lz4", "snappy", "zstd"])
@pytest.mark.parametrize("chunk,total", [(65536, 1000003), (131072, 131072 * 9), (4096, 70000), (262144, 100)])
def test_gpu_streams_decode_everywhere(eng, ref, ref_mt, codec, chunk, total):
    gc, oc = CODECS[codec]
    if codec == "zstd" and chunk > 131072:
        chunk = 131072
    data = a.corpus.silesia_mix(total, seed=5, segment=50000).tobytes()
    src = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(eng.device)
    stream_t = eng.rap_compress(gc, src, chunk)
    stream = bytes(stream_t.cpu().numpy())
    nparts = -(-total // chunk)
    skip = ref_mt.skip(stream)
    if nparts == 1:
        assert skip == 0 and stream == ol.Oracle().compress(oc, data)         # one partition: the plain one-shot stream
    elif codec == "zstd":
        assert skip == 0 and stream[:4] == bytes.fromhex("502a4d18")          # RAP frame inside a skippable frame
        ents = _entries(stream, 8)
        assert len(ents) == nparts and sum(e[2] for e in ents) == total
    else:
        ents = _entries(stream)
        assert skip == 16 + 12 * nparts and sum(e[2] for e in ents) == total
        assert ents[-1][0] + ents[-1][1] == len(stream)
    # legacy single-threaded decoder (zstd: skippable frames are part of the format)
    assert ref.decompress(oc, stream[skip:], total) == data
    # the reference's own multi-threaded decoder: one thread per partition (threads/threads.c:217-232 falls back to the
    # single-threaded decoder when there are fewer threads than partitions), and that fallback
    for threads in ((nparts, 3) if 1 < nparts <= 32 else (3,)):
        ref_mt.set_threads(threads)
        assert ref_mt.decompress(oc, stream, total) == data
    # GPU decoder
    dst, r = eng.rap_decompress(gc, stream_t, total)
    assert r == total and torch.equal(dst[:total], src)


@pytest.mark.parametrize("codec", ["lz4", "snappy", "zstd"])
def test_gpu_decodes_reference_mt_streams(eng, ref_mt, codec):
    gc, oc = CODECS[codec]
    for threads, total in ((2, 700001), (7, 3 << 20), (5, 2000000)):
        data = a.corpus.silesia_mix(total, seed=threads, segment=100000).tobytes()
        ref_mt.set_threads(threads)
        stream = ref_mt.compress(oc, data, _mt_cap(total, 64))
        assert stream is not None
        dst, r = eng.rap_decompress(gc, torch.frombuffer(bytearray(stream), dtype=torch.uint8).to(eng.device), total)
        assert r == total and bytes(dst[:total].cpu().numpy()) == data


def test_malformed_rap_streams_are_rejected(eng, ref_mt):
    data = a.corpus.make("text", 4 * 300000, seed=2).tobytes()
    ref_mt.set_threads(4)
    good = ref_mt.compress(ol.LZ4, data, _mt_cap(len(data), 4))
    for mutate in (lambda b: b[:len(b) - 1000],                       # truncated payload
                   lambda b: b[:20] + struct.pack("<I", 1 << 30) + b[24:],   # partition length out of range
                   lambda b: b[:70000] + bytes(64) + b[70064:]):     # corrupted sequences
        bad = mutate(good)
        dst, r = eng.rap_decompress(a.LZ4, torch.frombuffer(bytearray(bad), dtype=torch.uint8).to(eng.device), len(data))
        assert r < 0 or bytes(dst[:r].cpu().numpy()) != data


@pytest.mark.parametrize("codec", ["lz4", "snappy", "zstd"])
@pytest.mark.parametrize("kind,chunk,total", [("text", 4096, 100001), ("random", 1000, 40000), ("jsonlog", 65536, 65536 * 5 + 17),
                                              ("binary", 13, 700), ("database", 100, 5000), ("image16", 30000, 400000)])
def test_gpu_pack_equals_the_join_model(eng, oracle, ref, ref_mt, codec, kind, chunk, total):
    """Geometries the threaded reference cannot be steered into (tiny / ragged / all-literal partitions): the GPU join
    against the Python restatement of the reference's join (tests/rap_model.py, itself pinned to the threaded build)."""
    import rap_model
    gc, oc = CODECS[codec]
    data = a.corpus.make(kind, total, seed=chunk).tobytes()
    src = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(eng.device)
    stream_t = eng.rap_compress(gc, src, chunk)
    stream = bytes(stream_t.cpu().numpy())
    assert stream == rap_model.pack(oracle, oc, data, chunk)
    assert ref.decompress(oc, stream[ref_mt.skip(stream):], total) == data
    dst, r = eng.rap_decompress(gc, stream_t, total)
    assert r == total and torch.equal(dst[:total], src)


def _rank_segments(eng, gc, src, chunk, ranks):
    """what BASELINE config 5 leaves on every rank after the exchange: each 'rank' compresses its contiguous chunk range
    into its own slots, compacts them, and the compacted segments are laid out one after the other (16-byte aligned
    starts) in one buffer; returns (gathered tensor, chunk_off, chunk_len, chunk_src_len, chunk_aux)"""
    from aocl_compression_amd import sharding
    n = int(src.numel())
    nchunks = max(1, -(-n // chunk))
    segs, offs, lens, slens, auxs, base = [], [], [], [], [], 0
    for r in range(ranks):
        lo, hi = sharding.chunk_range(nchunks, r, ranks)
        if lo == hi:
            continue
        part = src[lo * chunk:min(n, hi * chunk)]
        slot = (a.compress_bound(gc, chunk) + 15) & ~15
        k = hi - lo
        slots = torch.empty(k * slot, dtype=torch.uint8, device=eng.device)
        out_len = torch.zeros(k, dtype=torch.int32, device=eng.device)
        aux = torch.zeros(k, dtype=torch.int32, device=eng.device)
        b = eng.uniform_compress_batch(part, chunk, slots, out_len, slot)
        b.aux = aux.data_ptr()
        eng._check(eng._fn(gc, "compress")(C.byref(b), eng._stream()), "compress")
        packed, off = eng.compact(slots, slot, out_len)
        segs.append(packed)
        o = off.cpu().numpy()[:k].astype(np.uint64) + base
        offs.append(o)
        lens.append(out_len.cpu().numpy())
        slens.append(np.array([min(chunk, part.numel() - i * chunk) for i in range(k)], np.uint32))
        auxs.append(aux.cpu().numpy().astype(np.uint32))
        base += (int(packed.numel()) + 15) & ~15
        if packed.numel() % 16:
            segs.append(torch.zeros(16 - packed.numel() % 16, dtype=torch.uint8, device=eng.device))
    return torch.cat(segs), np.concatenate(offs), np.concatenate(lens), np.concatenate(slens), np.concatenate(auxs)


@pytest.mark.parametrize("codec", ["lz4", "snappy", "zstd"])
@pytest.mark.parametrize("kind,chunk,total,ranks", [("text", 65536, 1000003, 2), ("jsonlog", 131072, 131072 * 9, 3),
                                                     ("random", 4096, 70000, 4), ("binary", 262144, 262144 * 5 + 77, 2)])
def test_assembled_stream_equals_the_packed_one(eng, ref, ref_mt, codec, kind, chunk, total, ranks):
    """aoclgpu_rap_assemble over chunks compressed 'elsewhere' (separate launches, separate slot buffers, only the compacted
    streams + sizes + aux travel) gives the bytes aoclgpu_rap_pack gives on one device, the reference decodes them (single-
    threaded after the frame skip, and with its MT decoder), and so does the GPU"""
    gc, oc = CODECS[codec]
    if codec == "zstd" and chunk > 131072:
        pytest.skip("zstd chunks are <= 128 KiB")
    data = a.corpus.make(kind, total, seed=17).tobytes()
    src = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(eng.device)
    want = bytes(eng.rap_compress(gc, src, chunk).cpu().numpy())
    gathered, off, ln, sl, ax = _rank_segments(eng, gc, src, chunk, ranks)
    got = bytes(eng.rap_assemble(gc, gathered, off, ln, sl, ax if codec == "lz4" else None).cpu().numpy())
    assert got == want
    skip = ref_mt.skip(got)
    if codec != "zstd":
        assert ref.decompress(oc, got[skip:], len(data)) == data
    assert ref_mt.decompress(oc, got, len(data)) == data
    dst, r = eng.rap_decompress(gc, torch.frombuffer(bytearray(got), dtype=torch.uint8).to(eng.device), len(data))
    assert r == len(data) and bytes(dst[:r].cpu().numpy()) == data


def test_assemble_all_literal_partitions_carry_across_ranks(eng, ref, ref_mt):
    """LZ4: partitions that are all literals travel on to the next partition's first token (lz4.c:2808-2823); with the chunks
    of several 'ranks' the carried run is stitched from the tails of several compressed chunks"""
    rng = np.random.default_rng(5)
    chunk = 13                                  # below LZ4_minLength: every chunk is one literal run
    data = bytes(rng.integers(0, 256, chunk * 7, dtype=np.uint8)) + (b"abcdabcdabcdabcdabcdabcd" * 40)
    src = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(eng.device)
    want = bytes(eng.rap_compress(a.LZ4, src, chunk).cpu().numpy())
    gathered, off, ln, sl, ax = _rank_segments(eng, a.LZ4, src, chunk, 3)
    got = bytes(eng.rap_assemble(a.LZ4, gathered, off, ln, sl, ax).cpu().numpy())
    assert got == want
    assert ref.decompress(ol.LZ4, got[ref_mt.skip(got):], len(data)) == data


def test_host_trim_releases_the_rap_host_cache(eng):
    """the process-wide buffer set of the RAP host calls must go back to the device on aoclgpu_host_trim (which
    aocl_llc_destroy calls), and a call above the keep limit must not leave its buffers behind."""
    import ctypes as C
    import numpy as np
    import aocl_compression_amd as a
    lib = eng.lib
    n = 8 << 20
    data = a.corpus.silesia_mix(n, seed=11)
    cap = a.compress_bound(a.LZ4, n) + 4096 + (n >> 8)
    out = np.empty(cap, np.uint8)
    r = lib.aoclgpu_rap_compress_host_ex(a.LZ4, data.ctypes.data, n, out.ctypes.data, cap, 0, 0)
    assert r > 0
    held = lib.aoclgpu_host_cache_bytes()
    assert held >= n, held                                     # kept for the next call of this size ...
    lib.aoclgpu_host_trim()
    assert lib.aoclgpu_host_cache_bytes() == 0                 # ... and released on request
    r2 = lib.aoclgpu_rap_compress_host_ex(a.LZ4, data.ctypes.data, n, out.ctypes.data, cap, 0, 0)
    assert r2 == r
    back = np.empty(n, np.uint8)
    lib.aoclgpu_rap_decompress_host.restype = C.c_int64
    lib.aoclgpu_rap_decompress_host.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    assert lib.aoclgpu_rap_decompress_host(a.LZ4, out.ctypes.data, r2, back.ctypes.data, n) == n and np.array_equal(back, data)
    lib.aoclgpu_host_trim()
    assert lib.aoclgpu_host_cache_bytes() == 0
