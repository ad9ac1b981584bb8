"""GPU parity: HIP LZ4 kernels (through the C-ABI) vs the CPU oracle -- bit-exact."""
import os

import numpy as np
import pytest
import torch

import aocl_compression_amd as a
import oracle_lib as ol

pytestmark = pytest.mark.gpu

KNOWN = [  # (input, reference output) -- from the compiled reference
    (b"", "00"), (b"a" * 13, "13610100506161616161"), (b"a" * 100, "1f6101004b506161616161"),
    (b"abcd" * 8, "4f61626364040004506461626364"), (b"hello world!", "c068656c6c6f20776f726c6421"),
]


def _chunks(raw, chunk):
    return [raw[i:i + chunk] for i in range(0, max(len(raw), 1), chunk)]


def _compress_and_check(eng, oracle, data: np.ndarray, chunk: int):
    src = torch.from_numpy(data).to(eng.device)
    slots, out_len, slot = eng.compress_chunks(a.LZ4, src, chunk)
    lens = out_len.cpu().tolist()
    host = slots.cpu().numpy()
    raw = data.tobytes()
    parts = _chunks(raw, chunk)
    assert len(lens) == len(parts)
    for i, part in enumerate(parts):
        want = oracle.compress(ol.LZ4, part)
        got = host[i * slot:i * slot + lens[i]].tobytes()
        assert lens[i] == len(want), (i, lens[i], len(want))
        assert got == want, f"chunk {i} differs"
    return src, slots, out_len, slot


def test_known_answers(eng):
    for inp, hexout in KNOWN:
        assert eng.compress_host(a.LZ4, inp).hex() == hexout
        assert eng.decompress_host(a.LZ4, bytes.fromhex(hexout), len(inp)) == inp


@pytest.mark.parametrize("kind", list(a.corpus.KINDS))
@pytest.mark.parametrize("chunk", [1000, 4096, 65536, 262144])
def test_compress_bit_exact(eng, oracle, kind, chunk):
    data = a.corpus.make(kind, (2 << 20) + 777, seed=11)
    src, slots, out_len, slot = _compress_and_check(eng, oracle, data, chunk)
    dst, dl = eng.decompress_chunks(a.LZ4, slots, out_len, slot, len(data), chunk)
    assert torch.equal(dst, src)
    exp = [len(c) for c in _chunks(data.tobytes(), chunk)]
    assert dl.cpu().tolist() == exp


@pytest.mark.parametrize("size", [1, 5, 12, 13, 14, 63, 64, 65, 4096, 65535, 65546, 65547, 65548, 100001, 1 << 20])
def test_ragged_sizes(eng, oracle, size):
    data = a.corpus.make("text", size, seed=size)
    raw = data.tobytes()
    got = eng.compress_host(a.LZ4, raw)
    assert got == oracle.compress(ol.LZ4, raw)
    assert eng.decompress_host(a.LZ4, got, size) == raw


def test_special_inputs(eng, oracle):
    rng = np.random.default_rng(5)
    cases = [
        bytes(300000), b"\xff" * 70000, bytes(range(256)) * 1200, b"ab" * 50000, b"abc" * 40000,
        rng.integers(0, 256, 300000, dtype=np.uint8).tobytes(),
        rng.integers(0, 2, 300000, dtype=np.uint8).tobytes(),
        (b"x" * 100 + rng.integers(0, 256, 5000, dtype=np.uint8).tobytes()) * 40,
    ]
    for raw in cases:
        got = eng.compress_host(a.LZ4, raw)
        assert got == oracle.compress(ol.LZ4, raw)
        assert eng.decompress_host(a.LZ4, got, len(raw)) == raw


def test_limited_output(eng, oracle):
    raw = a.corpus.make("jsonlog", 65536, seed=3).tobytes()
    full = oracle.compress(ol.LZ4, raw)
    for cap in [len(full) + 20, len(full) + 1, len(full), len(full) - 1, len(full) // 2, 16, 1, 0]:
        want = oracle.compress(ol.LZ4, raw, cap)
        got = eng.compress_host(a.LZ4, raw, cap)
        assert got == want, cap


def test_decode_foreign_and_malformed(eng, oracle):
    rng = np.random.default_rng(9)
    raw = a.corpus.make("binary", 200000, seed=4).tobytes()
    comp = oracle.compress(ol.LZ4, raw)
    assert eng.decompress_host(a.LZ4, comp, len(raw)) == raw
    assert eng.decompress_host(a.LZ4, comp, len(raw) - 1) is None      # output too small
    assert eng.decompress_host(a.LZ4, comp[:-3], len(raw)) is None     # truncated
    assert eng.decompress_host(a.LZ4, b"\x10\x61\x05\x00", 64) is None  # offset before start of block
    n_checked = 0
    for it in range(200):
        c = bytearray(comp[:4000] if it % 2 else comp)
        for _ in range(1 + it % 3):
            c[int(rng.integers(0, len(c)))] = int(rng.integers(0, 256))
        want = oracle.decompress(ol.LZ4, bytes(c), len(raw))
        got = eng.decompress_host(a.LZ4, bytes(c), len(raw))
        assert got == want
        n_checked += 1
    assert n_checked == 200


def test_decode_group_path_regressions(eng, oracle):
    """Inputs that drive the decoder's group path through its rare steps (round 4): long literal runs and matches between
    groups, groups that overflow the 2 KB block, and the fuzz input on which a refill in flight overwrote the part of the
    input ring the token chain had been rewound to (tools/fuzz_campaign.py seed 13759)."""
    import test_lz4_group_model as tm
    cases = [open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lz4_decode_rewind_seed13759_chunk2.bin"), "rb").read(),
             tm._lz_text(np.random.default_rng(5), 1500, 600, 4, 600)[:200000],
             tm._lz_text(np.random.default_rng(6), 3000, 20, 40, 200)[:250000]]
    for raw in cases:
        comp = oracle.compress(ol.LZ4, raw)
        assert eng.decompress_host(a.LZ4, comp, len(raw)) == raw
        assert eng.compress_host(a.LZ4, raw) == comp


def test_explicit_batch_ragged(eng, oracle):
    """Explicit (offset/length array) addressing with empty and ragged chunks."""
    import ctypes as C
    data = a.corpus.make("text", 500000, seed=21)
    raw = data.tobytes()
    lens = [0, 1, 13, 70000, 0, 262144, 65547, 12, 100000]
    offs = np.cumsum([0] + lens[:-1]).astype(np.uint64)
    caps = [a.compress_bound(a.LZ4, n) for n in lens]
    doffs = np.cumsum([0] + caps[:-1]).astype(np.uint64)
    t = torch
    dev = eng.device
    src = t.from_numpy(data).to(dev)
    dst = t.zeros(int(sum(caps)), dtype=t.uint8, device=dev)
    out_len = t.zeros(len(lens), dtype=t.int32, device=dev)
    d_so = t.from_numpy(offs.view(np.int64)).to(dev)
    d_sl = t.tensor(lens, dtype=t.int32, device=dev)
    d_do = t.from_numpy(doffs.view(np.int64)).to(dev)
    d_dc = t.tensor(caps, dtype=t.int32, device=dev)
    b = a.Batch(src=src.data_ptr(), dst=dst.data_ptr(), src_off=d_so.data_ptr(), src_len=d_sl.data_ptr(),
                dst_off=d_do.data_ptr(), dst_cap=d_dc.data_ptr(), out_len=out_len.data_ptr(), nchunks=len(lens))
    rc = eng.lib.aoclgpu_lz4_compress_chunks(C.byref(b), None)
    assert rc == 0
    t.cuda.synchronize()
    got_len = out_len.cpu().tolist()
    host = dst.cpu().numpy()
    for i, n in enumerate(lens):
        want = oracle.compress(ol.LZ4, raw[int(offs[i]):int(offs[i]) + n])
        assert host[int(doffs[i]):int(doffs[i]) + got_len[i]].tobytes() == want, i


def test_compaction(eng):
    data = a.corpus.make("jsonlog", 3 << 20, seed=2)
    src = torch.from_numpy(data).to(eng.device)
    slots, out_len, slot = eng.compress_chunks(a.LZ4, src, 262144)
    packed, off = eng.compact(slots, slot, out_len)
    lens = out_len.cpu().tolist()
    offs = off.cpu().tolist()
    assert offs[0] == 0 and offs[-1] == sum(lens) == packed.numel()
    hs, hp = slots.cpu().numpy(), packed.cpu().numpy()
    for i, ln in enumerate(lens):
        assert offs[i + 1] - offs[i] == ln
        assert np.array_equal(hp[offs[i]:offs[i] + ln], hs[i * slot:i * slot + ln])


def test_llc_api(eng, oracle):
    """aocl_llc_compress/decompress on host buffers keep the reference's contract (api/api.cpp:45-124)."""
    raw = a.corpus.make("text", 65536, seed=1).tobytes()      # BASELINE config 1
    r, out, d = eng.llc_compress(a.LZ4, raw, a.compress_bound(a.LZ4, len(raw)), stats=True)
    assert r == len(out) and out == oracle.compress(ol.LZ4, raw)
    assert d.cSize == r and d.cTime > 0 and d.cSpeed > 0
    r2, back, d2 = eng.llc_decompress(a.LZ4, out, len(raw), stats=True)
    assert r2 == len(raw) and back == raw and d2.dSize == len(raw)
    assert eng.llc_compress(a.LZ4, raw, 100)[0] == -2          # ERR_COMPRESSION_FAILED
    assert eng.llc_decompress(a.LZ4, out[:100], len(raw))[0] == -2
    assert eng.llc_compress(a.LZ4, b"", 16)[0] == 1            # api_gtest.cpp:638-673: inSize 0 -> 1


def test_full_size_round_trip(eng):
    """BASELINE config 2 size: 8 GiB of 256 KiB chunks -- round trip + per-chunk size sanity on device."""
    base = torch.from_numpy(a.corpus.silesia_mix(48 << 20, seed=0xA0C1)).to(eng.device)
    total = 8 << 30
    reps = -(-total // base.numel())
    src = base.repeat(reps)[:total].contiguous()
    chunk = 262144
    slots, out_len, slot = eng.compress_chunks(a.LZ4, src, chunk)
    assert int((out_len <= 0).sum().item()) == 0
    dst, dl = eng.decompress_chunks(a.LZ4, slots, out_len, slot, total, chunk)
    assert int((dl != chunk).sum().item()) == 0
    assert torch.equal(dst, src)
    # identical chunks (the base repeats every 48 MiB = 192 chunks) must compress identically
    per = (48 << 20) // chunk
    assert torch.equal(out_len[:per], out_len[per:2 * per])


def test_lz4hc_streams_decode(eng, ref):
    """LZ4HC: its streams are LZ4 blocks decoded by LZ4_decompress_safe (api/codec.cpp:179-188).  Streams
    from the reference's LZ4_compress_HC at several levels decode on the GPU; the unified API decodes them under the
    LZ4HC method and refuses to compress with it (no GPU match finder for the HC levels)."""
    import ctypes as C
    from aocl_compression_amd import api
    lib = api.load_library()
    ref.L.LZ4_compress_HC.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    for kind, size, level in (("text", 300000, 9), ("jsonlog", 65536, 1), ("binary", 1 << 20, 12), ("database", 5000, 4)):
        raw = a.corpus.make(kind, size, seed=level).tobytes()
        out = C.create_string_buffer(size + size // 255 + 16)
        r = ref.L.LZ4_compress_HC(raw, out, size, len(out), level)
        assert r > 0
        assert eng.decompress_host(a.LZ4, out.raw[:r], size) == raw
        d = api.LlcDesc()
        back = C.create_string_buffer(size)
        d.inBuf = C.cast(out, C.c_void_p); d.inSize = r; d.outBuf = C.cast(back, C.c_void_p); d.outSize = size
        assert lib.aocl_llc_setup(C.byref(d), 1) == 0
        assert lib.aocl_llc_decompress(C.byref(d), 1) == size and back.raw == raw
        d.inBuf = C.cast(back, C.c_void_p); d.inSize = size; d.outBuf = C.cast(out, C.c_void_p); d.outSize = len(out)
        assert lib.aocl_llc_compress(C.byref(d), 1) == -2                 # ERR_COMPRESSION_FAILED


@pytest.mark.parametrize("acc", [2, 3, 7, 64, 65, 1000, 65537])
def test_acceleration_bit_exact(eng, oracle, acc):
    """LZ4_compress_fast's acceleration on the GPU (AOCLGPU_LZ4_ACCELERATION in the batch flags, LZ4_compress_fast of the
    native API): bytes identical to the oracle for both table types and ragged chunks, streams decode back"""
    import ctypes as C
    oracle.L.orc_lz4_compress_fast.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int]
    for kind, chunk in (("text", 65536), ("binary", 262144), ("jsonlog", 5000), ("random", 70000), ("database", 1 << 20)):
        data = a.corpus.make(kind, (1 << 20) + 4321, seed=acc)
        src = torch.from_numpy(data).to(eng.device)
        slots, out_len, slot = eng.compress_chunks(a.LZ4, src, chunk, flags=a.api.lz4_acceleration(acc))
        lens = out_len.cpu().tolist(); host = slots.cpu().numpy(); raw = data.tobytes()
        for i, ln in enumerate(lens):
            part = raw[i * chunk:(i + 1) * chunk]
            cap = ol.bound(ol.LZ4, len(part))
            w = C.create_string_buffer(cap)
            r = oracle.L.orc_lz4_compress_fast(part, len(part), w, cap, acc)
            assert ln == r and host[i * slot:i * slot + ln].tobytes() == w.raw[:r], (kind, chunk, i)
        dst, dl = eng.decompress_chunks(a.LZ4, slots, out_len, slot, len(raw), chunk)
        assert torch.equal(dst, src)
    lib = eng.lib
    lib.LZ4_compress_fast.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int]
    part = a.corpus.make("text", 100000, seed=acc + 1).tobytes()
    cap = ol.bound(ol.LZ4, len(part))
    g = C.create_string_buffer(cap); w = C.create_string_buffer(cap)
    assert lib.LZ4_compress_fast(part, g, len(part), cap, acc) == oracle.L.orc_lz4_compress_fast(part, len(part), w, cap, acc)
    assert g.raw == w.raw


def _aliasing_inputs():
    """Inputs for the packed table's 17-bit entries (round 5): a position and the one 2^17 bytes behind it share an entry value,
    candidates further than 65535 bytes back must be rejected as the reference rejects them, and the age sweeps must have
    neutralised every entry before it turns 2^17 old - also across matches that jump over tens of KiB at once."""
    rng = np.random.default_rng(505)
    n = 262144
    out = {}
    # the same 2 KiB of text at distances of exactly / around 2^16 and 2^17
    for name, gaps in (("repeat_64k", (65535, 65536, 65537)), ("repeat_128k", (131071, 131072, 131073)), ("repeat_both", (65536, 65536, 65536))):
        buf = rng.integers(0, 256, n, dtype=np.uint8)
        blk = a.corpus.make("text", 2048, seed=3)
        pos = 1000
        buf[pos:pos + 2048] = blk
        for g in gaps:
            pos += g
            if pos + 2048 <= n:
                buf[pos:pos + 2048] = blk
        out[name] = buf
    # compressible data in front of and behind a run the match finder crosses in ONE step (100 KiB of zeros, then the same text again)
    t = a.corpus.make("text", n, seed=5)
    z = t.copy(); z[60000:60000 + 102400] = 0; z[60000 + 102400:] = t[:n - 60000 - 102400]
    out["jump_100k"] = z
    z2 = t.copy(); z2[1000:1000 + 200000] = 7
    out["jump_200k"] = z2
    # incompressible: long searches, batches of the schedule that span KiB
    out["random"] = rng.integers(0, 256, n, dtype=np.uint8)
    # records that repeat with a period just below / above the distance limit
    for per in (65530, 65540, 131060, 131080):
        rec = a.corpus.make("binary", per, seed=per)
        out[f"period_{per}"] = np.tile(rec, n // per + 1)[:n].copy()
    # a sparse page: few table writes per byte, slots outlive 2^16 positions
    sp = np.zeros(n, np.uint8); idx = rng.integers(0, n - 8, 3000); 
    for i in idx: sp[i:i + 8] = rng.integers(1, 256, 8, dtype=np.uint8)
    out["sparse"] = sp
    return out


@pytest.mark.parametrize("size", [131072, 131073, 196608, 200001, 262144])
def test_packed_table_aliasing_and_sweeps(eng, oracle, size):
    for name, buf in _aliasing_inputs().items():
        raw = buf[:size].tobytes()
        got = eng.compress_host(a.LZ4, raw)
        want = oracle.compress(ol.LZ4, raw)
        assert got == want, (name, size, len(got), len(want))
        assert eng.decompress_host(a.LZ4, got, len(raw)) == raw


def test_packed_table_batch_of_aliasing_chunks(eng, oracle):
    """the same inputs as chunks of one batch (256 KiB and a ragged tail), so that chunks in every sweep state run side by side"""
    bufs = list(_aliasing_inputs().values())
    data = np.concatenate(bufs + [bufs[0][:77777]])
    _compress_and_check(eng, oracle, data, 262144)


@pytest.mark.parametrize("jump", [59990, 60001, 63470, 65279, 65535, 65536, 65537, 65600, 65791, 65792, 65800, 70000, 131072])
def test_packed_table_long_jumps(eng, oracle, jump):
    """One match that jumps `jump` bytes (a period of 1993), followed at once by data that repeats what lies 65535 / 65534 / 40000
    bytes behind the jump's end - candidates written BEFORE the jump that are still (or just no longer) in range behind it: the
    age sweep of the one-at-a-time path must keep exactly those (fuzz campaign seed 19130, round 5)."""
    n = 262144
    for back in (65535, 65534, 40000):
        buf = a.corpus.make("text", n, seed=jump ^ back).copy()
        s0 = 74055
        e = min(s0 + jump, n - 6000)
        for i in range(s0, e):
            buf[i] = buf[i - 1993]
        if e - back >= 0:
            buf[e:e + 4537] = buf[e - back:e - back + 4537]
        raw = buf.tobytes()
        got = eng.compress_host(a.LZ4, raw)
        want = oracle.compress(ol.LZ4, raw)
        assert got == want, (jump, back, len(got), len(want))


@pytest.mark.parametrize("acc", [1, 5])
def test_packed_table_explicit_batch_and_accelerations(eng, oracle, acc):
    """the aliasing inputs through the per-chunk dispatching kernel (explicit batch), plain and with an acceleration - accelerated
    chunks above 128 KiB must take the u32 table (the packed table's sweeps are timed for acceleration 1)"""
    import ctypes as C
    ins = _aliasing_inputs()
    parts = [ins["jump_100k"].tobytes(), ins["period_131080"][:200001].tobytes(), ins["random"].tobytes(), ins["repeat_128k"].tobytes(),
             ins["sparse"][:140000].tobytes(), ins["jump_200k"].tobytes()]
    lens = [len(p) for p in parts]
    raw = b"".join(parts)
    offs = np.cumsum([0] + lens[:-1]).astype(np.uint64)
    caps = [a.compress_bound(a.LZ4, n) for n in lens]
    doffs = np.cumsum([0] + caps[:-1]).astype(np.uint64)
    t = torch; dev = eng.device
    src = t.frombuffer(bytearray(raw), dtype=t.uint8).to(dev)
    dst = t.zeros(int(sum(caps)), dtype=t.uint8, device=dev)
    out_len = t.zeros(len(lens), dtype=t.int32, device=dev)
    d_so = t.from_numpy(offs.view(np.int64)).to(dev); d_sl = t.tensor(lens, dtype=t.int32, device=dev)
    d_do = t.from_numpy(doffs.view(np.int64)).to(dev); d_dc = t.tensor(caps, dtype=t.int32, device=dev)
    b = a.Batch(src=src.data_ptr(), dst=dst.data_ptr(), src_off=d_so.data_ptr(), src_len=d_sl.data_ptr(),
                dst_off=d_do.data_ptr(), dst_cap=d_dc.data_ptr(), out_len=out_len.data_ptr(), nchunks=len(lens))
    b.reserved = a.api.lz4_acceleration(acc)
    assert eng.lib.aoclgpu_lz4_compress_chunks(C.byref(b), None) == 0
    t.cuda.synchronize()
    got_len = out_len.cpu().tolist(); host = dst.cpu().numpy()
    L = oracle.L
    L.orc_lz4_compress_fast.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int]
    for i, p in enumerate(parts):
        w = C.create_string_buffer(caps[i])
        r = L.orc_lz4_compress_fast(p, len(p), w, caps[i], acc)
        assert host[int(doffs[i]):int(doffs[i]) + got_len[i]].tobytes() == w.raw[:r], (acc, i)
    # the same chunks as a uniform batch with the acceleration (mode_for picks the table per launch)
    for p in (parts[0], parts[2]):
        s = t.frombuffer(bytearray(p), dtype=t.uint8).to(dev)
        slots, ol_, slot = eng.compress_chunks(a.LZ4, s, 262144, flags=a.api.lz4_acceleration(acc))
        w = C.create_string_buffer(a.compress_bound(a.LZ4, len(p)))
        r = L.orc_lz4_compress_fast(p, len(p), w, len(w), acc)
        assert slots[:int(ol_[0])].cpu().numpy().tobytes() == w.raw[:r], acc
