"""Pins the C oracle against the REAL reference compiled in place (oracle/_ref).  Skipped where neither the
prebuilt library nor /root/reference exists."""
import random

import numpy as np
import pytest

import aocl_compression_amd as a
import oracle_lib as ol
import zstd_frame_cases as zfc

CODECS = [ol.LZ4, ol.SNAPPY, ol.ZSTD]
SIZES = [0, 1, 6, 7, 12, 13, 14, 15, 16, 19, 20, 21, 63, 64, 65, 255, 256, 257, 1000, 1024, 4096, 16383, 16384, 16385, 40000,
         65535, 65536, 65546, 65547, 65548, 100000, 131071, 131072]


@pytest.mark.parametrize("kind", list(a.corpus.KINDS))
def test_compress_bit_exact(oracle, ref, kind):
    d = a.corpus.make(kind, 1 << 20, seed=31).tobytes()
    for codec in CODECS:
        for size in SIZES + ([262144, 1 << 20] if codec != ol.ZSTD else []):
            for off in (0, 4321):
                raw = d[off:off + size]
                got, want = oracle.compress(codec, raw), ref.compress(codec, raw)
                assert got == want, (codec, size, off)
                assert oracle.decompress(codec, got, max(len(raw), 1)) == raw


@pytest.mark.parametrize("kind", list(a.corpus.KINDS))
def test_zstd_level2_bit_exact(oracle, ref, kind):
    """level 2 is still the "fast" strategy for inputs of at most 128 KiB (2^15-slot table, minMatch 5 / 4)"""
    d = a.corpus.make(kind, 1 << 20, seed=37).tobytes()
    for size in SIZES:
        for off in (0, 4321, 777777):
            raw = d[off:off + size]
            got, want = oracle.compress(ol.ZSTD, raw, level=2), ref.compress(ol.ZSTD, raw, level=2)
            assert got == want, (size, off)
            assert oracle.decompress(ol.ZSTD, got, max(len(raw), 1)) == raw


def test_special_inputs(oracle, ref):
    rng = np.random.default_rng(5)
    cases = [bytes(131072), b"\xff" * 70000, bytes(range(256)) * 512, b"ab" * 50000, b"abc" * 40000,
             rng.integers(0, 256, 131072, dtype=np.uint8).tobytes(), rng.integers(0, 2, 131072, dtype=np.uint8).tobytes(),
             rng.integers(0, 16, 131072, dtype=np.uint8).tobytes(),
             rng.integers(0, 256, 40942, dtype=np.uint8).tobytes() + a.corpus.make("text", 40942, seed=8).tobytes()]
    for raw in cases:
        for codec in CODECS:
            assert oracle.compress(codec, raw) == ref.compress(codec, raw)


def test_lz4_limited_output(oracle, ref):
    raw = a.corpus.make("jsonlog", 65536, seed=3).tobytes()
    full = ref.compress(ol.LZ4, raw)
    for cap in [len(full) + 20, len(full) + 1, len(full), len(full) - 1, len(full) // 2, 16, 1, 0]:
        assert oracle.compress(ol.LZ4, raw, cap) == ref.compress(ol.LZ4, raw, cap), cap


def test_random_slices_zstd(oracle, ref):
    rng = random.Random(7)
    pools = {k: a.corpus.make(k, 2 << 20, seed=99).tobytes() for k in a.corpus.KINDS}
    for it in range(300):
        k = rng.choice(list(pools))
        size = rng.choice([rng.randrange(1, 300), rng.randrange(300, 20000), rng.randrange(20000, 131073)])
        off = rng.randrange(0, len(pools[k]) - size)
        raw = pools[k][off:off + size]
        if rng.random() < 0.2:
            k2 = rng.choice(list(pools))
            raw = raw[:size // 2] + pools[k2][off:off + size - size // 2]
        assert oracle.compress(ol.ZSTD, raw) == ref.compress(ol.ZSTD, raw), (k, size, off)


@pytest.mark.parametrize("codec", CODECS)
def test_decoder_on_mutated_streams(oracle, ref, codec):
    """Whatever the oracle accepts, the reference accepts with identical bytes (the oracle is at least as strict)."""
    rng = random.Random(codec)
    stricter = 0
    for it in range(400):
        n = rng.choice([50, 300, 5000, 70000])
        base = a.corpus.make(rng.choice(["text", "jsonlog", "binary"]), n, seed=it).tobytes()
        c = bytearray(ref.compress(codec, base))
        for _ in range(rng.randint(1, 3)):
            c[rng.randrange(len(c))] = rng.randrange(256)
        if rng.random() < 0.2:
            c = c[:rng.randrange(1, len(c) + 1)]
        cap = n + 64
        x, y = oracle.decompress(codec, bytes(c), cap), ref.decompress(codec, bytes(c), cap)
        if x is not None:
            assert y == x, it
        elif y is not None:
            stricter += 1
    if codec != ol.ZSTD:
        assert stricter == 0     # LZ4 / Snappy: identical accept set on this fuzz corpus


def test_decoder_on_mutated_multi_block_zstd_frames(oracle, ref):
    """Multi-block frames (small windows: dozens of blocks, repeat modes, RLE and raw blocks) mutated, truncated, extended, with a
    capacity a few bytes off: the oracle never accepts what the reference rejects and never decodes other bytes; it may only be
    stricter - and since round 6 only where a damaged Huffman stream is not consumed exactly (the reference's fast loops do not
    check that): the reserved bits of the sequence-modes byte, RLE / raw blocks above the frame's block size and the empty input
    are accepted as the reference accepts them (found by tools/fuzz_decode_frames_campaign.py seed 6134 after the GPU decoder
    had been aligned with the reference's error codes)."""
    import ctypes as C
    f = oracle.L.orc_zstd_compress_frame_cp; f.restype = C.c_size_t
    f.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int] + [C.c_uint] * 5
    rng = np.random.default_rng(6134)
    stricter = total = 0
    for it in range(120):
        kind = ["text", "mixed", "low", "rep", "zeros", "jsonlog", "random"][int(rng.integers(0, 7))]
        n = int(rng.choice([rng.integers(1, 5000), rng.integers(5000, 100000), rng.integers(131073, 200000)]))
        raw = zfc.make(kind, n, int(rng.integers(0, 1 << 30)))
        out = C.create_string_buffer(n + (n >> 8) + 64)
        r = f(raw, n, out, n + (n >> 8) + 64, int(rng.integers(1, 3)), int(rng.choice([10, 12, 14, 17, 19])), int(rng.integers(7, 16)), int(rng.integers(7, 16)),
              int(rng.integers(4, 8)), 0)
        assert r > 0
        for _ in range(4):
            comp = bytearray(out.raw[:r]); cap = n
            m = int(rng.integers(0, 8))
            if m <= 3:
                for _ in range(m if m else 1):
                    comp[int(rng.integers(0, len(comp)))] = int(rng.integers(0, 256))
            elif m == 4 and len(comp) > 2:
                comp = comp[:int(rng.integers(1, len(comp)))]
            elif m == 5:
                comp += bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))
            elif m == 6:
                cap = max(0, n + int(rng.integers(-5, 6)))
            x, y = oracle.decompress(ol.ZSTD, bytes(comp), cap), ref.decompress(ol.ZSTD, bytes(comp), cap)
            total += 1
            if x is not None:
                assert y == x, it
            elif y is not None:
                stricter += 1
    assert stricter <= total // 10, (stricter, total)
    assert oracle.decompress(ol.ZSTD, b"", 16) == ref.decompress(ol.ZSTD, b"", 16) == b""
    # one sequence-modes byte with its reserved bits set: both accept
    raw = zfc.make("text", 3000, 5)
    c = bytearray(ref.compress(ol.ZSTD, raw))
    for i in range(9, len(c)):                       # find a byte whose low two bits the reference ignores: flip them everywhere, keep what it accepts
        d = bytearray(c); d[i] ^= 3
        y = ref.decompress(ol.ZSTD, bytes(d), len(raw))
        if y == raw:
            assert oracle.decompress(ol.ZSTD, bytes(d), len(raw)) == raw, i


def test_decoder_never_returns_other_bytes_than_the_reference(oracle, ref):
    """Small blocks of skewed alphabets (Huffman streams of a few bytes, sequence sections of a few dozen) with one or two damaged
    bytes, a third of them zeroed: the shapes that found, in round 6, (a) that the reference's four-stream Huffman loops make no
    end-of-stream check (accepted now when bits are LEFT OVER and the streams are long enough for those loops, refused when a
    stream runs out of bits) and (b) that a sequence bitstream that runs out of bits wraps around the reference's bit container
    (refused now; the zeros read there before could give OTHER bytes than the reference's with a success value).  The oracle may
    refuse what the reference accepts - never the other way round, never other bytes."""
    rng = np.random.default_rng(5)
    stricter = total = 0
    for it in range(700):
        nsym = int(rng.choice([2, 3, 4, 6, 10, 30, 100, 256]))
        n = int(rng.choice([rng.integers(256, 600), rng.integers(600, 3000), rng.integers(3000, 40000)]))
        p = rng.dirichlet(np.ones(nsym) * rng.choice([0.2, 1.0, 5.0]))
        raw = rng.choice(nsym, size=n, p=p).astype(np.uint8).tobytes()
        comp0 = ref.compress(ol.ZSTD, raw)
        for mu in range(4):
            comp = bytearray(comp0)
            for _ in range(int(rng.integers(1, 3))):
                i = int(rng.integers(min(8, len(comp) - 1), len(comp)))
                comp[i] = int(rng.integers(0, 256)) if rng.random() < 0.7 else 0
            x, y = oracle.decompress(ol.ZSTD, bytes(comp), n), ref.decompress(ol.ZSTD, bytes(comp), n)
            total += 1
            if x is not None:
                assert y == x, (it, mu)
            elif y is not None:
                stricter += 1
    assert stricter <= total // 5, (stricter, total)


def test_decode_foreign_zstd_frames(oracle, ref):
    import ctypes as C
    rng = random.Random(3)
    pool = a.corpus.make("text", 1 << 20, seed=6).tobytes()
    ref.L.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    for it in range(40):
        size = rng.choice([0, 1, 100, 5000, 70000, 131072, 300000, 600000])
        off = rng.randrange(0, len(pool) - size)
        raw = pool[off:off + size]
        out = C.create_string_buffer(len(raw) + len(raw) // 200 + 1024)
        r = ref.L.ZSTD_compress(out, len(out), raw, len(raw), rng.choice([1, 3, 5, 9, 19, -3]))
        assert oracle.decompress(ol.ZSTD, out.raw[:r], max(len(raw), 1)) == raw


def test_lz4_acceleration_bit_exact(oracle, ref):
    """LZ4_compress_fast(.., acceleration): the probe step starts at `acceleration` (lz4.c:1486-1510); out-of-range values are
    clamped to 1..65537 (lz4.c:2521-2522).  Both table types (below / above 65547 bytes), limited output included."""
    import ctypes as C
    ref.L.LZ4_compress_fast.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int]
    oracle.L.orc_lz4_compress_fast.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int]
    for kind in ("text", "binary", "random", "jsonlog"):
        for n in (0, 13, 100, 5000, 65546, 65547, 300000):
            data = a.corpus.make(kind, n, seed=31).tobytes() if n else b""
            cap = ol.bound(ol.LZ4, n)
            for acc in (-3, 0, 1, 2, 3, 7, 64, 65, 1000, 65537, 70000):
                for c in (cap, max(1, cap // 3)):
                    w = C.create_string_buffer(cap + 8); g = C.create_string_buffer(cap + 8)
                    rw = ref.L.LZ4_compress_fast(data, w, n, c, acc)
                    rg = oracle.L.orc_lz4_compress_fast(data, n, g, c, acc)
                    assert rw == rg and w.raw[:max(rw, 0)] == g.raw[:max(rg, 0)], (kind, n, acc, c)


def _ref_cparams(ref, level, n):
    import ctypes as C
    class CP(C.Structure):
        _fields_ = [(k, C.c_uint) for k in ("windowLog", "chainLog", "hashLog", "searchLog", "minMatch", "targetLength")] + [("strategy", C.c_int)]
    ref.L.ZSTD_getCParams.restype = CP
    ref.L.ZSTD_getCParams.argtypes = [C.c_int, C.c_ulonglong, C.c_size_t]
    return ref.L.ZSTD_getCParams(level, n, 0)


@pytest.mark.parametrize("kind", list(a.corpus.KINDS))
def test_zstd_double_fast_bit_exact(oracle, ref, kind):
    """levels 3 and 4 (double-fast, zstd_double_fast.c) on inputs of at most 128 KiB: the oracle's parse with the reference's
    own parameters for (level, size) gives the reference's frame.  The reference's one-shot ZSTD_compress runs its dispatcher
    first, which selects the AOCL level table (zstd_compress.c:7390-7425); aocl_setup_zstd_encode does the same up front so
    that ZSTD_getCParams reports the rows ZSTD_compress will use."""
    import ctypes as C
    ref.L.aocl_setup_zstd_encode.restype = C.c_void_p
    ref.L.aocl_setup_zstd_encode.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t]
    ref.L.aocl_setup_zstd_encode(0, 3, 0, 3, 0)
    oracle.L.orc_zstd_compress_cparams.restype = C.c_size_t
    oracle.L.orc_zstd_compress_cparams.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int, C.c_uint, C.c_uint, C.c_uint]
    seen = set()
    for n in (0, 1, 6, 7, 20, 27, 28, 40, 100, 513, 1000, 4096, 16384, 16385, 30000, 65536, 100000, 131071, 131072):
        data = a.corpus.make(kind, n, seed=41 + n).tobytes() if n else b""
        for level in (3, 4):
            cp = _ref_cparams(ref, level, n)
            if cp.strategy != 2:
                continue                                       # (level 4 is "greedy" for inputs of at most 16 KiB)
            want = ref.compress(ol.ZSTD, data, level=level)
            cap = ol.bound(ol.ZSTD, n)
            out = C.create_string_buffer(max(cap, 16))
            r = oracle.L.orc_zstd_compress_cparams(data, n, out, max(cap, 16), 2, cp.hashLog, cp.chainLog, cp.minMatch)
            assert want is not None and out.raw[:r] == want, (kind, n, level, cp.hashLog, cp.chainLog, cp.minMatch)
            assert oracle.decompress(ol.ZSTD, want, n) == data
            seen.add((cp.hashLog, cp.chainLog, cp.minMatch))
    assert len(seen) >= 4


@pytest.mark.parametrize("kind", zfc.KINDS)
def test_zstd_multi_block_frames_bit_exact(oracle, ref, kind):
    """Inputs above 128 KiB: one frame of 128 KiB blocks (min(128 KiB, window) in general) that share the hash table, the
    repeat offsets and the last compressed block's Huffman table (ZSTD_compress_frameChunk, zstd_compress.c:4504-4583;
    HUF_compress_internal's table reuse, huf_compress.c:1386-1432; RLE blocks after the first, :4372-4383).  The level's own
    parameters and explicit ones: a window below the input (window descriptor, ZSTD_window_enforceMaxDist), a window below
    the block size (smaller blocks), every minMatch of the fast strategy."""
    import ctypes as C
    f = oracle.L.orc_zstd_compress_frame_tl
    f.restype = C.c_size_t
    f.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_uint, C.c_uint, C.c_uint, C.c_uint]
    checked = 0
    for i, n in enumerate(zfc.EDGE_SIZES + (3000000,)):
        data = zfc.make(kind, n, seed=100 + i)
        cps = [zfc.ref_level_cparams(ref, 1, n)]
        cps += [zfc.CParams(*row) for row in ((17, 13, 14, 1, 6, 0, 1), (12, 13, 13, 1, 5, 0, 1), (19, 13, 14, 1, 7, 0, 1), (23, 13, 12, 1, 4, 0, 1),
                                                (19, 12, 13, 1, 6, 1, 1), (11, 12, 9, 1, 3, 1, 1))]   # targetLength 1: raw literals; minMatch 3 = 4
        l2 = zfc.ref_level_cparams(ref, 2, n)
        if l2.strategy == 1:
            cps.append(l2)
        for cp in cps:
            want = zfc.ref_frame(ref, data, cp)
            cap = n + (n >> 8) + 64
            out = C.create_string_buffer(cap)
            r = f(data, n, out, cap, cp.windowLog, cp.hashLog, max(cp.minMatch, 4), cp.targetLength)
            assert out.raw[:r] == want, (kind, n, cp.windowLog, cp.hashLog, cp.minMatch, cp.targetLength)
            checked += 1
        if i % 4 == 0:
            assert oracle.decompress(ol.ZSTD, want, n) == data
    assert checked >= 50


@pytest.mark.parametrize("kind", zfc.KINDS)
def test_zstd_fast_with_a_wider_step_bit_exact(oracle, ref, kind):
    """the levels below -1 (and explicit targetLength > 1): the fast parser's pairs are targetLength + 1 bytes apart from the start
    (zstd_fast.c:440, :491-506), literals stay raw (zstd_compress_internal.h:609-622) - the reference's level rows and explicit sets"""
    import ctypes as C
    f = oracle.L.orc_zstd_compress_frame_tl
    f.restype = C.c_size_t
    f.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_uint, C.c_uint, C.c_uint, C.c_uint]
    checked = 0
    for i, n in enumerate((1, 700, 20000, 131072, 131073, 400000, 1300000)):
        data = zfc.make(kind, n, seed=900 + i)
        cps = [zfc.ref_level_cparams(ref, level, n) for level in (-2, -3, -5, -7, -64, -1000, -131072)]
        cps += [zfc.CParams(*row) for row in ((17, 13, 14, 1, 6, 2, 1), (19, 13, 12, 1, 4, 9, 1), (20, 13, 17, 1, 7, 300, 1))]
        for cp in cps:
            assert cp.strategy == 1 and cp.targetLength > 1
            want = zfc.ref_frame(ref, data, cp)
            cap = n + (n >> 8) + 64
            out = C.create_string_buffer(cap)
            r = f(data, n, out, cap, cp.windowLog, cp.hashLog, max(cp.minMatch, 4), cp.targetLength)
            assert out.raw[:r] == want, (kind, n, cp.windowLog, cp.hashLog, cp.minMatch, cp.targetLength)
            checked += 1
    assert checked >= 60


@pytest.mark.parametrize("kind", zfc.KINDS)
def test_zstd_double_fast_multi_block_frames_bit_exact(oracle, ref, kind):
    """the same for "double-fast" (zstd_double_fast.c:332-573): both tables, the window and the repeat offsets carry from block
    to block - the levels' own rows (level 2 between 128 and 256 KiB, levels 3-4) and explicit parameters, minMatch 3 = 4"""
    import ctypes as C
    f = oracle.L.orc_zstd_compress_frame_cp
    f.restype = C.c_size_t
    f.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int] + [C.c_uint] * 5
    checked = 0
    for i, n in enumerate(zfc.EDGE_SIZES + (3000000, 100000)):
        data = zfc.make(kind, n, seed=500 + i)
        cps = [cp for cp in (zfc.ref_level_cparams(ref, level, n) for level in (2, 3, 4)) if cp.strategy == 2]
        cps += [zfc.CParams(*row) for row in ((17, 15, 16, 1, 5, 0, 2), (12, 10, 11, 1, 4, 0, 2), (21, 16, 17, 1, 7, 0, 2), (18, 12, 12, 1, 3, 0, 2))]
        for cp in cps:
            want = zfc.ref_frame(ref, data, cp)
            cap = n + (n >> 8) + 64
            out = C.create_string_buffer(cap)
            r = f(data, n, out, cap, 2, cp.windowLog, cp.hashLog, cp.chainLog, max(cp.minMatch, 4), cp.targetLength)
            assert out.raw[:r] == want, (kind, n, cp.windowLog, cp.hashLog, cp.chainLog, cp.minMatch)
            checked += 1
    assert checked >= 60


@pytest.mark.parametrize("kind", zfc.KINDS)
def test_lz4_linked_blocks_bit_exact(oracle, ref, kind):
    """the blocks of LZ4F_compressFrame with linked blocks (the frame format's default): orc_lz4_compress_linked_blocks restates
    LZ4_compress_fast_continue in prefix mode (lz4.c:3179-3223, :972-1360) block after block; the reference's own frame is cut
    at its block headers and compared block by block (stored blocks included), three accelerations"""
    import ctypes as C
    import os
    import struct
    import lz4f_lib as lf
    if not hasattr(ref.L, "LZ4F_compressFrame"):
        pytest.skip("reference build without lz4frame.c")
    R = lf.bind(ref.L)
    f = oracle.L.orc_lz4_compress_linked_blocks
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32)]
    BS = {4: 65536, 5: 262144, 6: 1 << 20, 7: 4 << 20}
    cases = ((4, 65537), (4, 200000), (4, 1 << 20), (5, 262145), (5, 900000), (6, (1 << 20) + 77), (6, 3500000), (7, (4 << 20) + 5),
             (4, 65536 * 3), (4, 65536 * 2 + 12), (4, 65536 + 5))
    for i, (bsid, n) in enumerate(cases):
        data = zfc.make(kind, n, seed=800 + i)
        for level in (0, -1, -5):
            frame = lf.compress(R, data, lf.prefs(bsid=bsid, level=level))
            assert isinstance(frame, bytes) and not (frame[4] >> 5) & 1
            pos, blocks = 7, []
            while True:
                (h,) = struct.unpack_from("<I", frame, pos)
                pos += 4
                if h == 0:
                    break
                blocks.append((h, frame[pos:pos + (h & 0x7FFFFFFF)]))
                pos += h & 0x7FFFFFFF
            out = C.create_string_buffer(n + 16)
            lens = (C.c_uint32 * (n // BS[bsid] + 2))()
            nb = f(data, n, BS[bsid], -level + 1 if level < 0 else 1, out, n + 16, lens)
            assert nb == len(blocks), (kind, bsid, n, level)
            at = 0
            for k in range(nb):
                sz = lens[k] & 0x7FFFFFFF
                assert lens[k] == blocks[k][0] and out.raw[at:at + sz] == blocks[k][1], (kind, bsid, n, level, k)
                at += sz


def test_huffman_sort_bucket_of_count_164(oracle, ref):
    """nine or more literal symbols with a count of exactly 164: the reference's HUF_sort permutes them (zstd_frame_cases.
    count164_case); found by tools/fuzz_frames_campaign.py on the GPU, pinned here for the oracle"""
    for seed in range(12):
        data = zfc.count164_case(seed, ties=9 + seed)
        assert oracle.compress(ol.ZSTD, data) == ref.compress(ol.ZSTD, data), seed
