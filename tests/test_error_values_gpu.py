"""Exact RETURN VALUES of the decoders on malformed input, against the real reference (oracle/_ref): the bar for integer work is
bit-exact, return values included (verdict r05 item 7).
  LZ4_decompress_safe   -(ip - src) - 1 at the reference's point of failure (lz4.c:3799-3802), the size otherwise
  ZSTD_decompress       (size_t)-ZSTD_error_xxx (zstd_errors.h:60-106): compared as the raw size_t, which is what ZSTD_getErrorCode decodes
  snappy_uncompress     SNAPPY_OK / SNAPPY_INVALID_INPUT / SNAPPY_BUFFER_TOO_SMALL (snappy-c.cc:34-88)
Deviations that are deliberate are listed in INTEGRATION.md and pinned here (KNOWN_*)."""
import ctypes as C

import numpy as np
import pytest

import aocl_compression_amd as a
from aocl_compression_amd import api
import oracle_lib as ol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib(eng):
    L = api.load_library()
    L.LZ4_decompress_safe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.ZSTD_decompress.restype = C.c_size_t
    L.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.snappy_uncompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t)]
    return L


@pytest.fixture(scope="module")
def ref():
    if not ol.have_ref():
        pytest.skip("oracle/_ref/libaocl_ref.so not built")
    r = ol.Ref()
    r.L.LZ4_decompress_safe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    return r


def _mutations(comp: bytes, rng, count, short=4000):
    yield comp, "intact"
    for cut in (1, 2, 3, 5, 17, len(comp) // 2):
        if cut < len(comp):
            yield comp[:-cut], f"truncated by {cut}"
    for it in range(count):
        c = bytearray(comp[:short] if it % 2 else comp)
        for _ in range(1 + it % 3):
            c[int(rng.integers(0, len(c)))] = int(rng.integers(0, 256))
        yield bytes(c), f"mutation {it}"


def test_lz4_decompress_safe_returns_the_references_value(lib, ref):
    rng = np.random.default_rng(61)
    bad = []
    n = 0
    for kind, size in (("binary", 200000), ("text", 70000), ("jsonlog", 262144), ("database", 5000)):
        raw = a.corpus.make(kind, size, seed=11).tobytes()
        comp = ref.compress(ol.LZ4, raw)
        for c, what in _mutations(comp, rng, 120):
            for cap in (len(raw), len(raw) - 1, len(raw) // 2):
                w = C.create_string_buffer(max(cap, 1)); g = C.create_string_buffer(max(cap, 1))
                want = ref.L.LZ4_decompress_safe(c, w, len(c), cap)
                got = lib.LZ4_decompress_safe(c, g, len(c), cap)
                n += 1
                if want != got or (want > 0 and w.raw[:want] != g.raw[:want]):
                    bad.append((kind, what, cap, want, got))
    assert not bad, f"{len(bad)} of {n} return values differ from the reference's, e.g. {bad[:8]}"


def ERR(code):
    return (1 << 64) - code


def test_zstd_decompress_returns_the_references_error_code(lib, ref):
    rng = np.random.default_rng(62)
    bad = []
    n = strict = 0
    for kind, size in (("binary", 131072), ("text", 70000), ("jsonlog", 100000), ("random", 3000), ("database", 5000)):
        raw = a.corpus.make(kind, size, seed=12).tobytes()
        comp = ref.compress(ol.ZSTD, raw)
        for c, what in _mutations(comp, rng, 100, short=600):
            for cap in (len(raw), len(raw) - 1):
                w = C.create_string_buffer(max(cap, 1)); g = C.create_string_buffer(max(cap, 1))
                want = ref.L.ZSTD_decompress(w, cap, c, len(c))
                got = lib.ZSTD_decompress(g, cap, c, len(c))
                n += 1
                if want == got and (want > cap or w.raw[:want] == g.raw[:want]):
                    continue
                # The ONE deliberate kind of deviation ("Error VALUES"): a Huffman stream or a sequence bitstream that RUNS
                # OUT of bits is corruption_detected here; the reference's bit readers wrap around / read on in front of the stream and
                # return what that decodes to (success, or dstSize_tooSmall when the output then runs out).  Never the other way round,
                # and never different bytes with a success value.
                if got == ERR(20) and (want <= cap or want == ERR(70)):
                    strict += 1
                    continue
                bad.append((kind, what, cap, hex(want), hex(got)))
    assert not bad, f"{len(bad)} of {n} return values differ from the reference's, e.g. {bad[:8]}"
    assert strict <= n // 8, f"{strict} of {n} inputs rejected where the reference decodes a damaged literal stream: more than the mutations explain"


def test_snappy_uncompress_returns_the_references_status(lib, ref):
    rng = np.random.default_rng(63)
    bad = []
    n = 0
    for kind, size in (("jsonlog", 65536), ("text", 70000), ("binary", 200000), ("database", 3000)):
        raw = a.corpus.make(kind, size, seed=13).tobytes()
        comp = ref.compress(ol.SNAPPY, raw)
        for c, what in _mutations(comp, rng, 120):
            for cap in (len(raw), len(raw) - 1, 10):
                w = C.create_string_buffer(max(cap, 1)); g = C.create_string_buffer(max(cap, 1))
                wl = C.c_size_t(cap); gl = C.c_size_t(cap)
                want = ref.L.snappy_uncompress(c, len(c), w, C.byref(wl))
                got = lib.snappy_uncompress(c, len(c), g, C.byref(gl))
                n += 1
                if want != got or (want == 0 and (wl.value != gl.value or w.raw[:wl.value] != g.raw[:gl.value])):
                    bad.append((kind, what, cap, want, got))
    assert not bad, f"{len(bad)} of {n} statuses differ from the reference's, e.g. {bad[:8]}"


def test_zstd_damaged_small_blocks_never_other_bytes(lib, ref):
    """The adversarial shape of tests/test_oracle_vs_ref.py (small blocks of skewed alphabets, one or two damaged bytes, a third of
    them zeroed) through the GPU decoder against the reference itself: same value, or corruption_detected where the reference's
    bit readers wrap around a stream that ran out of bits - never success with other bytes, never success where the reference fails."""
    rng = np.random.default_rng(15)
    bad = []
    n_calls = strict = 0
    for it in range(500):
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
            c = bytes(comp)
            w = C.create_string_buffer(n); g = C.create_string_buffer(n)
            want = ref.L.ZSTD_decompress(w, n, c, len(c)); got = lib.ZSTD_decompress(g, n, c, len(c))
            n_calls += 1
            if want == got and (want > n or w.raw[:want] == g.raw[:want]):
                continue
            if got == ERR(20) and (want <= n or want == ERR(70)):
                strict += 1
                continue
            bad.append((it, mu, n, hex(want), hex(got)))
    assert not bad, f"{len(bad)} of {n_calls}: {bad[:8]}"
    assert strict <= n_calls // 5, (strict, n_calls)


def test_lz4_and_snappy_damaged_small_blocks(lib, ref):
    """The same adversarial shape for the two byte-oriented decoders: exact value / status and bytes, no exception."""
    rng = np.random.default_rng(16)
    bad = []
    for it in range(400):
        nsym = int(rng.choice([2, 4, 10, 100, 256]))
        n = int(rng.choice([rng.integers(20, 300), rng.integers(300, 3000), rng.integers(3000, 70000)]))
        p = rng.dirichlet(np.ones(nsym) * rng.choice([0.2, 1.0, 5.0]))
        raw = rng.choice(nsym, size=n, p=p).astype(np.uint8).tobytes()
        for codec in (ol.LZ4, ol.SNAPPY):
            comp0 = ref.compress(codec, raw)
            for mu in range(3):
                comp = bytearray(comp0)
                for _ in range(int(rng.integers(1, 3))):
                    i = int(rng.integers(0, len(comp)))
                    comp[i] = int(rng.integers(0, 256)) if rng.random() < 0.7 else 0
                c = bytes(comp)
                cap = n if mu else max(0, n - int(rng.integers(0, 4)))
                w = C.create_string_buffer(max(cap, 1)); g = C.create_string_buffer(max(cap, 1))
                if codec == ol.LZ4:
                    want = ref.L.LZ4_decompress_safe(c, w, len(c), cap); got = lib.LZ4_decompress_safe(c, g, len(c), cap)
                    ok = want == got and (want <= 0 or w.raw[:want] == g.raw[:want])
                else:
                    wl = C.c_size_t(cap); gl = C.c_size_t(cap)
                    want = ref.L.snappy_uncompress(c, len(c), w, C.byref(wl)); got = lib.snappy_uncompress(c, len(c), g, C.byref(gl))
                    ok = want == got and (want != 0 or (wl.value == gl.value and w.raw[:wl.value] == g.raw[:gl.value]))
                if not ok:
                    bad.append((codec, it, mu, n, cap, want, got))
    assert not bad, f"{len(bad)}: {bad[:8]}"


def test_zstd_streams_of_several_frames(lib, ref):
    """Concatenated frames (each with its content size: the host path decodes them side by side), skippable frames between them,
    a capacity that ends inside the second or third frame, damage in one of them, trailing bytes: the value of the FIRST failure in
    stream order, as the reference's frame loop reports it (zstd_decompress.c:1114-1197)."""
    import struct
    rng = np.random.default_rng(17)
    bad = []
    n_calls = strict = 0
    for it in range(150):
        parts = [a.corpus.make(["text", "binary", "jsonlog", "database"][int(rng.integers(0, 4))], int(rng.integers(1, 60000)), seed=it * 7 + k).tobytes()
                 for k in range(int(rng.integers(2, 5)))]
        frames = [ref.compress(ol.ZSTD, p_) for p_ in parts]
        if rng.random() < 0.3:
            frames.insert(int(rng.integers(0, len(frames) + 1)), struct.pack("<II", 0x184D2A50 + int(rng.integers(0, 16)), 5) + b"skip!")
        stream = bytearray(b"".join(frames)); total = sum(len(p_) for p_ in parts)
        for mu in range(4):
            c = bytearray(stream); cap = total
            m = int(rng.integers(0, 6))
            if m == 0:
                cap = max(0, total - int(rng.integers(1, total + 1)))
            elif m == 1:
                c = c[:int(rng.integers(1, len(c)))]
            elif m == 2:
                c += bytes(rng.integers(0, 256, int(rng.integers(1, 12)), dtype=np.uint8))
            elif m <= 4:
                for _ in range(m - 2):
                    c[int(rng.integers(0, len(c)))] = int(rng.integers(0, 256))
            cb = bytes(c)
            w = C.create_string_buffer(max(cap, 1)); g = C.create_string_buffer(max(cap, 1))
            want = ref.L.ZSTD_decompress(w, cap, cb, len(cb)); got = lib.ZSTD_decompress(g, cap, cb, len(cb))
            n_calls += 1
            if want == got and (want > cap or w.raw[:want] == g.raw[:want]):
                continue
            if got == ERR(20) and (want <= cap or want == ERR(70)):
                strict += 1
                continue
            bad.append((it, mu, m, cap, total, hex(want), hex(got)))
    assert not bad, f"{len(bad)} of {n_calls}: {bad[:8]}"
    assert strict <= n_calls // 8, (strict, n_calls)
