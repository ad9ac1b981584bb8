"""GPU parity: HIP Zstd level-1 kernels (through the C-ABI) vs the CPU oracle -- bit-exact frames."""
import ctypes as C

import numpy as np
import pytest
import torch

import aocl_compression_amd as a
import oracle_lib as ol

pytestmark = pytest.mark.gpu

KNOWN = [(b"", "28b52ffd2000010000"), (b"a" * 100, "28b52ffd206445000010616101003f012c"),
         (b"abcd" * 8, "28b52ffd202055000020616263640100338e08")]   # SURVEY.md 8c, from the compiled reference


def _chunks(raw, chunk):
    return [raw[i:i + chunk] for i in range(0, max(len(raw), 1), chunk)]


def test_known_answers(eng):
    for inp, hexout in KNOWN:
        assert eng.compress_host(a.ZSTD, inp).hex() == hexout


@pytest.mark.parametrize("kind", list(a.corpus.KINDS))
@pytest.mark.parametrize("chunk", [131072, 65536, 16384, 4096])
def test_compress_bit_exact(eng, oracle, kind, chunk):
    data = a.corpus.make(kind, (1 << 20) + 555, seed=13)
    src = torch.from_numpy(data).to(eng.device)
    slots, out_len, slot = eng.compress_chunks(a.ZSTD, src, chunk)
    lens = out_len.cpu().tolist()
    host = slots.cpu().numpy()
    parts = _chunks(data.tobytes(), chunk)
    bad = []
    for i, part in enumerate(parts):
        want = oracle.compress(ol.ZSTD, part)
        got = host[i * slot:i * slot + lens[i]].tobytes()
        if got != want:
            bad.append((i, len(part), lens[i], len(want)))
    assert not bad, bad[:5]


@pytest.mark.parametrize("size", [1, 6, 7, 19, 20, 21, 63, 64, 65, 255, 256, 257, 1000, 1023, 1024, 4095, 16383, 16384,
                                  16385, 65535, 65791, 65792, 100001, 131071, 131072])
def test_ragged_sizes(eng, oracle, size):
    raw = a.corpus.make("text", size, seed=size).tobytes()
    assert eng.compress_host(a.ZSTD, raw) == oracle.compress(ol.ZSTD, raw)


def test_special_inputs(eng, oracle):
    rng = np.random.default_rng(5)
    cases = [
        bytes(131072), b"\xff" * 70000, bytes(range(256)) * 512, b"ab" * 50000, b"abc" * 40000,
        rng.integers(0, 256, 131072, dtype=np.uint8).tobytes(),
        rng.integers(0, 2, 131072, dtype=np.uint8).tobytes(),
        rng.integers(0, 16, 131072, dtype=np.uint8).tobytes(),
        (b"x" * 100 + rng.integers(0, 256, 3000, dtype=np.uint8).tobytes()) * 40,
        bytes(70000) + rng.integers(0, 256, 60000, dtype=np.uint8).tobytes(),
        rng.integers(0, 256, 66000, dtype=np.uint8).tobytes() + bytes(65000),
        rng.integers(0, 256, 40942, dtype=np.uint8).tobytes() + a.corpus.make("text", 40942, seed=8).tobytes(),
    ]
    for i, raw in enumerate(cases):
        assert eng.compress_host(a.ZSTD, raw) == oracle.compress(ol.ZSTD, raw), i


def test_small_output_buffer_fails(eng, oracle):
    raw = a.corpus.make("text", 5000, seed=3).tobytes()
    want = oracle.compress(ol.ZSTD, raw)
    assert eng.compress_host(a.ZSTD, raw, a.compress_bound(a.ZSTD, len(raw)) - 1) == want     # below the bound, but the frame fits
    assert eng.compress_host(a.ZSTD, raw, len(want) - 1) is None
    assert eng.compress_host(a.ZSTD, raw, 8) is None


@pytest.mark.parametrize("kind", list(a.corpus.KINDS))
@pytest.mark.parametrize("chunk", [131072, 16384])
def test_round_trip(eng, kind, chunk):
    data = a.corpus.make(kind, (1 << 20) + 555, seed=14)
    src = torch.from_numpy(data).to(eng.device)
    slots, out_len, slot = eng.compress_chunks(a.ZSTD, src, chunk)
    dst, dl = eng.decompress_chunks(a.ZSTD, slots, out_len, slot, len(data), chunk)
    assert torch.equal(dst, src)
    assert dl.cpu().tolist() == [len(p) for p in _chunks(data.tobytes(), chunk)]


def test_decode_reference_frames(eng, ref):
    """Frames produced by the real reference at several levels (multi-block, other strategies) decode on the GPU."""
    import ctypes as C
    rng = np.random.default_rng(3)
    pools = {k: a.corpus.make(k, 1 << 20, seed=6).tobytes() for k in ("text", "jsonlog", "binary", "database", "image16")}
    ref.L.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    for it in range(60):
        k = list(pools)[it % len(pools)]
        size = int(rng.choice([0, 1, 100, 5000, 70000, 131072, 300000, 600000]))
        off = int(rng.integers(0, len(pools[k]) - size))
        raw = pools[k][off:off + size]
        level = int(rng.choice([1, 1, 3, 5, 9, 19, -3]))
        out = C.create_string_buffer(len(raw) + len(raw) // 200 + 1024)
        r = ref.L.ZSTD_compress(out, len(out), raw, len(raw), level)
        assert eng.decompress_host(a.ZSTD, out.raw[:r], max(len(raw), 1)) == raw, (k, size, level)


def test_decode_malformed(eng, oracle):
    rng = np.random.default_rng(9)
    raw = a.corpus.make("text", 90000, seed=4).tobytes()
    comp = oracle.compress(ol.ZSTD, raw)
    assert eng.decompress_host(a.ZSTD, comp, len(raw)) == raw
    assert eng.decompress_host(a.ZSTD, comp, len(raw) - 1) is None
    assert eng.decompress_host(a.ZSTD, comp[:-3], len(raw)) is None
    assert eng.decompress_host(a.ZSTD, b"\x00\x01\x02\x03\x04", 64) is None
    for it in range(150):
        c = bytearray(comp)
        for _ in range(1 + it % 3):
            c[int(rng.integers(0, len(c)))] = int(rng.integers(0, 256))
        want = oracle.decompress(ol.ZSTD, bytes(c), len(raw))
        got = eng.decompress_host(a.ZSTD, bytes(c), len(raw))
        assert got == want, it


def test_llc_api(eng, oracle):
    raw = a.corpus.make("text", 131072, seed=1).tobytes()      # BASELINE config 4 block
    cap = a.compress_bound(a.ZSTD, len(raw))
    r, out, d = eng.llc_compress(a.ZSTD, raw, cap, level=1, stats=True)
    assert r == len(out) and out == oracle.compress(ol.ZSTD, raw) and d.cSize == r
    r2, back, _ = eng.llc_decompress(a.ZSTD, out, len(raw))
    assert r2 == len(raw) and back == raw
    assert eng.llc_decompress(a.ZSTD, out[:50], len(raw))[0] == -2
    assert eng.llc_compress(a.ZSTD, b"", 64)[0] == 9           # api_gtest.cpp:638-673: inSize 0 -> > 0


def test_full_size_round_trip(eng):
    """BASELINE config 4 size: 8 GiB of 128 KiB blocks -- round trip on device."""
    base = torch.from_numpy(a.corpus.silesia_mix(48 << 20, seed=0xA0C4)).to(eng.device)
    total = 8 << 30
    reps = -(-total // base.numel())
    src = base.repeat(reps)[:total].contiguous()
    chunk = 131072
    slots, out_len, slot = eng.compress_chunks(a.ZSTD, src, chunk)
    assert int((out_len <= 0).sum().item()) == 0
    dst, dl = eng.decompress_chunks(a.ZSTD, slots, out_len, slot, total, chunk)
    assert int((dl != chunk).sum().item()) == 0
    assert torch.equal(dst, src)


# ---- level 2 (the levels above 1 that still use the "fast" strategy at these sizes) ----
@pytest.mark.parametrize("kind", ["text", "jsonlog", "binary", "database", "random"])
@pytest.mark.parametrize("chunk", [131072, 50000, 16384, 4096])
def test_level2_bit_exact(eng, oracle, kind, chunk):
    data = a.corpus.make(kind, (1 << 19) + 777, seed=17)
    src = torch.from_numpy(data).to(eng.device)
    slots, out_len, slot = eng.compress_chunks(a.ZSTD, src, chunk, flags=a.api.FLAG_ZSTD_LEVEL2)
    lens = out_len.cpu().tolist()
    host = slots.cpu().numpy()
    parts = _chunks(data.tobytes(), chunk)
    bad = []
    for i, part in enumerate(parts):
        want = oracle.compress(ol.ZSTD, part, level=2)
        if host[i * slot:i * slot + lens[i]].tobytes() != want:
            bad.append((i, len(part), lens[i], len(want)))
    assert not bad, bad[:5]
    dst, dl = eng.decompress_chunks(a.ZSTD, slots, out_len, slot, len(data), chunk)
    assert torch.equal(dst, src)


@pytest.mark.parametrize("size", [1, 7, 20, 21, 255, 1000, 4096, 16384, 16385, 65536, 100001, 131072])
def test_level2_sizes_and_native_api(eng, oracle, ref, size):
    raw = a.corpus.make("text", size, seed=size + 1).tobytes()
    want = oracle.compress(ol.ZSTD, raw, level=2)
    assert eng.compress_host(a.ZSTD, raw, flags=a.api.FLAG_ZSTD_LEVEL2) == want
    cap = len(raw) + (len(raw) >> 8) + 1024
    out = C.create_string_buffer(cap)
    r = ref.L.ZSTD_compress(out, cap, raw, len(raw), 2)                  # the reference itself
    assert out.raw[:r] == want
    L = a.api.load_library()
    L.ZSTD_compress.restype = C.c_size_t
    L.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    out2 = C.create_string_buffer(cap)
    r2 = L.ZSTD_compress(out2, cap, raw, len(raw), 2)                    # the GPU library under the reference's name
    assert out2.raw[:r2] == want


class _CP(__import__("ctypes").Structure):
    _fields_ = [(k, __import__("ctypes").c_uint) for k in ("windowLog", "chainLog", "hashLog", "searchLog", "minMatch", "targetLength")] + [("strategy", __import__("ctypes").c_int)]


@pytest.mark.parametrize("upstream", [False, True])
@pytest.mark.parametrize("level", [3, 4])
def test_double_fast_levels_bit_exact(eng, oracle, level, upstream):
    """levels 3 and 4 (double-fast, zstd_double_fast.c): every chunk of a batch gets the parameters the reference derives
    for (level, chunk size) from the AOCL or the upstream level table; frames byte-identical to the oracle's parse with those
    parameters (pinned to the reference in tests/test_oracle_vs_ref.py), and they decode back"""
    import ctypes as C
    lib = eng.lib
    lib.ZSTD_getCParams.restype = _CP
    lib.ZSTD_getCParams.argtypes = [C.c_int, C.c_ulonglong, C.c_size_t]
    lib.aocl_setup_zstd_encode.restype = C.c_void_p
    lib.aocl_setup_zstd_encode.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t]
    oracle.L.orc_zstd_compress_cparams.restype = C.c_size_t
    oracle.L.orc_zstd_compress_cparams.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int, C.c_uint, C.c_uint, C.c_uint]
    lib.aocl_destroy_zstd_encode()                       # (a setup is ignored while the dispatcher counts as set up, as in the reference)
    lib.aocl_setup_zstd_encode(1 if upstream else 0, 0, 0, level, 0)
    try:
        for kind, chunk in (("text", 131072), ("binary", 65536), ("jsonlog", 20000), ("database", 100000), ("random", 40000),
                            ("text", 16384), ("binary", 5000), ("jsonlog", 700)):
            total = 6 * chunk + 777
            data = a.corpus.make(kind, total, seed=level * 7 + chunk)
            src = torch.from_numpy(data).to(eng.device)
            slots, out_len, slot = eng.compress_chunks(a.ZSTD, src, chunk, flags=a.api.zstd_level(level, upstream))
            lens = out_len.cpu().tolist(); host = slots.cpu().numpy(); raw = data.tobytes()
            for i, ln in enumerate(lens):
                part = raw[i * chunk:(i + 1) * chunk]
                cp = lib.ZSTD_getCParams(level, len(part), 0)
                if cp.strategy != 2:
                    assert ln == 0, (kind, chunk, i)                    # level 4 at <= 16 KiB is "greedy": no GPU parse, reported as 0
                    continue
                cap = ol.bound(ol.ZSTD, len(part))
                w = C.create_string_buffer(max(cap, 16))
                r = oracle.L.orc_zstd_compress_cparams(part, len(part), w, max(cap, 16), 2, cp.hashLog, cp.chainLog, cp.minMatch)
                assert ln == r and host[i * slot:i * slot + ln].tobytes() == w.raw[:r], (kind, chunk, i, level, upstream)
            if all(l > 0 for l in lens):
                dst, dl = eng.decompress_chunks(a.ZSTD, slots, out_len, slot, len(raw), chunk)
                assert torch.equal(dst, src)
    finally:
        lib.aocl_destroy_zstd_encode()
        lib.aocl_setup_zstd_encode(0, 0, 0, 1, 0)


def test_slots_smaller_than_the_bound(eng, oracle):
    """a destination smaller than ZSTD_compressBound succeeds whenever the frame itself fits (as ZSTD_compress does), and fails
    (0 / dstSize_tooSmall) one byte below that"""
    import ctypes as C
    lib = eng.lib
    lib.ZSTD_compress.restype = C.c_size_t
    lib.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    lib.ZSTD_isError.argtypes = [C.c_size_t]
    for kind, n in (("text", 100000), ("random", 50000), ("binary", 131072), ("text", 1), ("text", 300)):
        data = a.corpus.make(kind, n, seed=61).tobytes()
        want = oracle.compress(ol.ZSTD, data)
        assert eng.compress_host(a.ZSTD, data, cap=len(want)) == want
        assert eng.compress_host(a.ZSTD, data, cap=len(want) - 1) is None
        out = C.create_string_buffer(len(want))
        assert lib.ZSTD_compress(out, len(want), data, n, 1) == len(want) and out.raw == want
        assert lib.ZSTD_isError(lib.ZSTD_compress(out, len(want) - 1, data, n, 1))


def test_huffman_sort_bucket_of_count_164(eng, oracle):
    """the reference quicksorts the bucket of count 164 as well (huf_compress.c:649-664, one bucket below the log2 buckets): nine
    or more symbols with that count come out permuted, and with them their code lengths (found by tools/fuzz_frames_campaign.py)"""
    import zstd_frame_cases as zfc
    t = eng.torch
    for seed in range(24):
        data = zfc.count164_case(seed, ties=9 + seed % 14)
        src = t.frombuffer(bytearray(data), dtype=t.uint8).to(eng.device)
        slots, lens, slot = eng.compress_chunks(a.ZSTD, src, len(data))
        got = slots[:int(lens[0])].cpu().numpy().tobytes()
        assert got == oracle.compress(ol.ZSTD, data), seed


def test_search_that_starts_at_the_last_lane(eng, oracle):
    """tests/golden/zstd_walk_search_at_lane63.bin (tools/fuzz_campaign.py seed 9903; CPU twin in tests/test_zstd_chain_model.py):
    the search that follows a repcode match ending at lane 62 starts at the batch's last lane"""
    import os
    data = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "zstd_walk_search_at_lane63.bin"), "rb").read()
    t = eng.torch
    src = t.frombuffer(bytearray(data), dtype=t.uint8).to(eng.device)
    slots, lens, slot = eng.compress_chunks(a.ZSTD, src, len(data))
    assert slots[:int(lens[0])].cpu().numpy().tobytes() == oracle.compress(ol.ZSTD, data)


def test_double_fast_falls_back_when_the_slab_scratch_cannot_be_had(eng, oracle):
    """the two-kernel path needs 1 MB of scratch per chunk in flight; when the device cannot give it (here: the test hook
    AOCLGPU_FAULT_WORKSPACE_MB fails every request above 6 GB with the runtime's own out-of-memory error) the call falls back to
    the fused kernel, and the failed allocation does not surface as that launch's error"""
    import os
    import ctypes as C
    level, chunk, nchunks = 3, 65536, 8200
    lib = eng.lib
    lib.aocl_setup_zstd_encode.restype = C.c_void_p
    lib.aocl_setup_zstd_encode.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t]
    lib.aoclgpu_zstd_last_split_slabs.restype = C.c_int
    data = a.corpus.silesia_mix(chunk * nchunks, seed=9)
    src = torch.from_numpy(data).to(eng.device)
    lib.aocl_destroy_zstd_encode(); lib.aocl_setup_zstd_encode(0, 0, 0, level, 0)
    try:
        slots0, len0, slot = eng.compress_chunks(a.ZSTD, src, chunk, flags=a.api.zstd_level(level, False))
        assert lib.aoclgpu_zstd_last_split_slabs() >= 1                 # enough memory: finder + entropy kernels
        os.environ["AOCLGPU_FAULT_WORKSPACE_MB"] = "6000"
        side = torch.cuda.Stream()                                      # (another stream: its scratch has to be allocated)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            slots1, len1, _ = eng.compress_chunks(a.ZSTD, src, chunk, flags=a.api.zstd_level(level, False))
        side.synchronize()
        assert lib.aoclgpu_zstd_last_split_slabs() == 0                 # the fused kernel ran
        assert torch.equal(len0, len1) and int((len1 <= 0).sum()) == 0
        # same bytes either way (a slot's tail behind the chunk's compressed size is whatever the buffer held)
        pos = torch.arange(slots0.numel(), device=slots0.device)
        live = (pos % slot) < len1.to(torch.int64).repeat_interleave(slot)
        assert torch.equal(slots0[live], slots1[live])
    finally:
        os.environ.pop("AOCLGPU_FAULT_WORKSPACE_MB", None)
        lib.aocl_destroy_zstd_encode(); lib.aocl_setup_zstd_encode(0, 0, 0, 1, 0)


@pytest.mark.parametrize("slab_mb", [None, "2500"])
def test_double_fast_large_batch_two_kernels(eng, oracle, slab_mb):
    """more chunks than the fused kernel holds at once: the finder and the entropy stage run as two kernels over slabs of
    per-chunk workspaces (zstd_df_find_kernel / zstd_df_entropy_kernel); one slab, and two slabs with a small slab budget"""
    import os
    import ctypes as C
    level = 3
    lib = eng.lib
    lib.ZSTD_getCParams.restype = _CP
    lib.ZSTD_getCParams.argtypes = [C.c_int, C.c_ulonglong, C.c_size_t]
    lib.aocl_setup_zstd_encode.restype = C.c_void_p
    lib.aocl_setup_zstd_encode.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t]
    oracle.L.orc_zstd_compress_cparams.restype = C.c_size_t
    oracle.L.orc_zstd_compress_cparams.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int, C.c_uint, C.c_uint, C.c_uint]
    chunk, nchunks = 65536, 4300                        # (the two-kernel path is for chunks of 64 KiB and more)
    data = a.corpus.silesia_mix(chunk * nchunks - 777, seed=level).tobytes()
    lib.aocl_destroy_zstd_encode()
    lib.aocl_setup_zstd_encode(0, 0, 0, level, 0)
    old = os.environ.get("AOCLGPU_ZSTD_DF_SLAB_MB")
    try:
        if slab_mb:
            os.environ["AOCLGPU_ZSTD_DF_SLAB_MB"] = slab_mb
        src = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(eng.device)
        slots, out_len, slot = eng.compress_chunks(a.ZSTD, src, chunk, flags=a.api.zstd_level(level, False))
        lens = out_len.cpu().tolist(); host = slots.cpu().numpy()
        assert all(l > 0 for l in lens)
        for k in list(range(0, len(lens), 61)) + [len(lens) - 1]:
            part = data[k * chunk:(k + 1) * chunk]
            cp = lib.ZSTD_getCParams(level, len(part), 0)
            cap = max(ol.bound(ol.ZSTD, len(part)), 16); w = C.create_string_buffer(cap)
            r = oracle.L.orc_zstd_compress_cparams(part, len(part), w, cap, cp.strategy, cp.hashLog, cp.chainLog, cp.minMatch)
            assert host[k * slot:k * slot + lens[k]].tobytes() == w.raw[:r], k
        dst, dl = eng.decompress_chunks(a.ZSTD, slots, out_len, slot, len(data), chunk)
        assert torch.equal(dst, src)
    finally:
        if old is None:
            os.environ.pop("AOCLGPU_ZSTD_DF_SLAB_MB", None)
        else:
            os.environ["AOCLGPU_ZSTD_DF_SLAB_MB"] = old
        lib.aocl_destroy_zstd_encode()
        lib.aocl_setup_zstd_encode(0, 0, 0, 1, 0)
