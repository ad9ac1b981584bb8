"""GPU parity: HIP Snappy kernels (through the C-ABI) vs the CPU oracle -- bit-exact."""
import numpy as np
import pytest
import torch

import aocl_compression_amd as a
import oracle_lib as ol

pytestmark = pytest.mark.gpu

KNOWN = [(b"", "00"), (b"a", "010061"), (b"abc", "0308616263"), (b"a" * 100, "640061fe01008a0100"),
         (b"abcd" * 8, "200c616263646e0400")]   # SURVEY.md 8c, from the compiled reference


def _chunks(raw, chunk):
    return [raw[i:i + chunk] for i in range(0, max(len(raw), 1), chunk)]


@pytest.fixture(params=["auto", "hbm-tables", "mixed"], autouse=True)
def table_placement(request, eng):
    """every test runs three times: with the launcher's own choice (LDS tables for these small batches), with the global-table
    kernel pinned (AOCLGPU_FLAG_SNAPPY_TABLES_HBM) and with the mixed kernel pinned (.._MIXED), which otherwise only large
    batches reach"""
    eng.force_flags = {"auto": 0, "hbm-tables": a.api.FLAG_SNAPPY_TABLES_HBM,
                       "mixed": a.api.FLAG_SNAPPY_TABLES_MIXED}[request.param]   # mixed: one LDS-table wave + three HBM-table waves
    yield request.param
    eng.force_flags = 0


def test_known_answers(eng):
    for inp, hexout in KNOWN:
        assert eng.compress_host(a.SNAPPY, inp).hex() == hexout
        assert eng.decompress_host(a.SNAPPY, bytes.fromhex(hexout), max(len(inp), 1)) == inp


@pytest.mark.parametrize("kind", list(a.corpus.KINDS))
@pytest.mark.parametrize("chunk", [1000, 4096, 65536, 262144])
def test_compress_bit_exact(eng, oracle, kind, chunk):
    data = a.corpus.make(kind, (2 << 20) + 333, seed=12)
    src = torch.from_numpy(data).to(eng.device)
    slots, out_len, slot = eng.compress_chunks(a.SNAPPY, src, chunk)
    lens = out_len.cpu().tolist()
    host = slots.cpu().numpy()
    parts = _chunks(data.tobytes(), chunk)
    for i, part in enumerate(parts):
        want = oracle.compress(ol.SNAPPY, part)
        assert host[i * slot:i * slot + lens[i]].tobytes() == want, f"chunk {i}"
    dst, dl = eng.decompress_chunks(a.SNAPPY, slots, out_len, slot, len(data), chunk)
    assert torch.equal(dst, src)
    assert dl.cpu().tolist() == [len(p) for p in parts]


@pytest.mark.parametrize("size", [1, 14, 15, 16, 17, 60, 61, 255, 256, 257, 4095, 16384, 16385, 65535, 65536, 65537,
                                  100001, 1 << 20])
def test_ragged_sizes(eng, oracle, size):
    raw = a.corpus.make("jsonlog", size, seed=size).tobytes()
    got = eng.compress_host(a.SNAPPY, raw)
    assert got == oracle.compress(ol.SNAPPY, raw)
    assert eng.decompress_host(a.SNAPPY, got, size) == raw


def test_special_inputs(eng, oracle):
    rng = np.random.default_rng(5)
    cases = [
        bytes(300000), b"\xff" * 70000, bytes(range(256)) * 1200, b"ab" * 50000, b"abc" * 40000,
        rng.integers(0, 256, 300000, dtype=np.uint8).tobytes(),
        rng.integers(0, 2, 300000, dtype=np.uint8).tobytes(),
        (b"x" * 100 + rng.integers(0, 256, 5000, dtype=np.uint8).tobytes()) * 40,
    ]
    for raw in cases:
        got = eng.compress_host(a.SNAPPY, raw)
        assert got == oracle.compress(ol.SNAPPY, raw)
        assert eng.decompress_host(a.SNAPPY, got, len(raw)) == raw


def test_small_output_buffer_fails(eng):
    raw = a.corpus.make("text", 5000, seed=3).tobytes()
    assert eng.compress_host(a.SNAPPY, raw, a.compress_bound(a.SNAPPY, len(raw)) - 1) is None   # api/codec.cpp:271-274


def test_decode_handbuilt_and_malformed(eng, oracle):
    """Hand-built streams in the spirit of snappy_gtest.cpp:75-222,357-447 + random corruption."""
    def lit(b):
        n = len(b) - 1
        return (bytes([n << 2]) if n < 60 else bytes([60 << 2, n]) if n < 256 else bytes([61 << 2, n & 255, n >> 8])) + b
    def copy4(off, ln):
        return bytes([3 | ((ln - 1) << 2)]) + off.to_bytes(4, "little")
    good = bytes([20]) + lit(b"0123456789") + copy4(10, 10)            # 4-byte-offset copy
    assert eng.decompress_host(a.SNAPPY, good, 20) == b"0123456789" * 2 == oracle.decompress(ol.SNAPPY, good, 20)
    bad_streams = [
        bytes([10]) + lit(b"abc") + bytes([2 | (3 << 2), 0, 0]),           # zero offset copy
        bytes([0x80, 0x80, 0x80, 0x80, 0x80, 0x01]) + lit(b"abc"),         # varint too long
        bytes([0xff, 0xff, 0xff, 0xff, 0x7f]) + lit(b"abc"),               # varint overflows 32 bits
        bytes([0x80]),                                                     # unterminated varint
        bytes([5]) + lit(b"abc"),                                          # header lies (too long)
        bytes([2]) + lit(b"abc"),                                          # header lies (too short)
        bytes([10]) + lit(b"abc") + bytes([2 | (3 << 2), 9]),              # truncated copy tag
        bytes([10]) + lit(b"abc") + bytes([2 | (3 << 2), 9, 0]),           # offset beyond produced
        b"",
    ]
    for s in bad_streams:
        assert oracle.decompress(ol.SNAPPY, s, 64) is None
        assert eng.decompress_host(a.SNAPPY, s, 64) is None
    rng = np.random.default_rng(9)
    raw = a.corpus.make("binary", 150000, seed=4).tobytes()
    comp = oracle.compress(ol.SNAPPY, raw)
    for it in range(200):
        c = bytearray(comp[:3000] if it % 2 else comp)
        for _ in range(1 + it % 3):
            c[int(rng.integers(0, len(c)))] = int(rng.integers(0, 256))
        assert eng.decompress_host(a.SNAPPY, bytes(c), len(raw) + 64) == oracle.decompress(ol.SNAPPY, bytes(c), len(raw) + 64)


def test_llc_api(eng, oracle):
    raw = a.corpus.make("jsonlog", 65536, seed=1).tobytes()      # BASELINE config 3 block
    cap = a.compress_bound(a.SNAPPY, len(raw))
    r, out, d = eng.llc_compress(a.SNAPPY, raw, cap, stats=True)
    assert r == len(out) and out == oracle.compress(ol.SNAPPY, raw) and d.cSize == r
    r2, back, _ = eng.llc_decompress(a.SNAPPY, out, len(raw))
    assert r2 == len(raw) and back == raw
    assert eng.llc_compress(a.SNAPPY, raw, cap - 1)[0] == -2      # outSize < MaxCompressedLength
    assert eng.llc_decompress(a.SNAPPY, out, len(raw) - 1)[0] == -2
    assert eng.llc_compress(a.SNAPPY, b"", 32)[0] == 1            # api_gtest.cpp:638-673


def test_full_size_round_trip(eng):
    """BASELINE config 3 shape: JSON/log data in 64 KiB blocks, 8 GiB, round trip on device."""
    base = torch.from_numpy(a.corpus.make("jsonlog", 64 << 20, seed=0xA0C3)).to(eng.device)
    total = 8 << 30
    src = base.repeat(total // base.numel())
    chunk = 65536
    slots, out_len, slot = eng.compress_chunks(a.SNAPPY, src, chunk)
    assert int((out_len <= 0).sum().item()) == 0
    dst, dl = eng.decompress_chunks(a.SNAPPY, slots, out_len, slot, total, chunk)
    assert int((dl != chunk).sum().item()) == 0
    assert torch.equal(dst, src)


def test_large_batch_all_kernels_agree(eng, table_placement):
    """a batch big enough for the persistent kernels (8 192 fragments): LDS-table, HBM-table and mixed kernels give the same
    bytes, run after run (chunks are handed out dynamically, the result must not depend on who took which)"""
    if table_placement != "auto":
        pytest.skip("runs once")
    base = torch.from_numpy(a.corpus.make("jsonlog", 64 << 20, seed=3)).to(eng.device)
    src = base.repeat(8)
    outs = {}
    F = a.api
    for name, flag in (("lds", F.FLAG_SNAPPY_TABLES_LDS), ("hbm", F.FLAG_SNAPPY_TABLES_HBM), ("mixed", F.FLAG_SNAPPY_TABLES_MIXED),
                       ("mixed again", F.FLAG_SNAPPY_TABLES_MIXED), ("auto", 0)):
        slots, out_len, slot = eng.compress_chunks(a.SNAPPY, src, 65536, flags=flag)
        used = (torch.arange(slots.numel(), device=slots.device) % slot) < out_len.to(torch.int64).repeat_interleave(slot)
        outs[name] = (out_len.clone(), torch.where(used, slots, torch.zeros_like(slots)))
    ref_len, ref_bytes = outs["lds"]
    for name, (ln, by) in outs.items():
        assert torch.equal(ln, ref_len), name
        assert torch.equal(by, ref_bytes), name
