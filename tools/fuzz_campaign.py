#!/usr/bin/env python3
"""Random-seed parity campaign on the GPU box (TEST INFRASTRUCTURE: loads oracle/): LZ-grammar inputs as in
tests/test_fuzz_gpu.py, but unbounded in number and covering what the fixed-seed test does not - LZ4 accelerations, all three
LZ4 table modes (sizes around 64 KiB and 256 KiB), Zstd levels 1-4 with both level tables, ragged batches.  Every GPU output
must equal the oracle's byte for byte and decode back.  usage: tools/fuzz_campaign.py <seconds> [seed] [lz4|snappy|zstd]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import aocl_compression_amd as a
import oracle_lib as ol

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
only = sys.argv[3] if len(sys.argv) > 3 else None
eng = a.engine(0); orc = ol.Oracle(); L = orc.L; lib = eng.lib
L.orc_lz4_compress_fast.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int]
L.orc_zstd_compress_cparams.restype = C.c_size_t
L.orc_zstd_compress_cparams.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int, C.c_uint, C.c_uint, C.c_uint]


class CP(C.Structure):
    _fields_ = [(k, C.c_uint) for k in ("windowLog", "chainLog", "hashLog", "searchLog", "minMatch", "targetLength")] + [("strategy", C.c_int)]


lib.ZSTD_getCParams.restype = CP
lib.ZSTD_getCParams.argtypes = [C.c_int, C.c_ulonglong, C.c_size_t]
lib.aocl_setup_zstd_encode.restype = C.c_void_p
lib.aocl_setup_zstd_encode.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t]


def build(rng, target):
    out = bytearray()
    alphabet = int(rng.choice([2, 4, 16, 256]))
    while len(out) < target:
        k = rng.integers(0, 5)
        if k == 0:
            out += bytes(rng.integers(0, alphabet, int(rng.integers(1, 60)), dtype=np.uint8))
        elif k == 1 and out:
            d = 1 + int(rng.integers(0, min(len(out), 70000))); n = int(rng.integers(3, 400))
            if d >= n: out += out[len(out) - d:len(out) - d + n]
            else:
                for _ in range(n): out.append(out[-d])
        elif k == 2:
            out += bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 600))
        elif k == 3 and len(out) >= 9:
            p = int(rng.integers(1, 10)); n = int(rng.integers(4, 300))
            for _ in range(n): out.append(out[-p])
        elif k == 4 and len(out) > 64:      # a longer copy (long matches, specials)
            d = 1 + int(rng.integers(0, min(len(out), 65535))); n = int(rng.integers(30, 3000))
            if d >= n: out += out[len(out) - d:len(out) - d + n]
            else:
                seg = bytes(out[-d:]); out += (seg * (n // d + 1))[:n]
    return bytes(out[:target])


def seq_shaped(rng, n):
    """a prescribed parse: fresh random literals of a few fixed run lengths, each followed by a copy of a few fixed lengths at a
    few fixed distances (overlapping copies, distances 1..4, lengths at the formats' token boundaries: LZ4 15 / 19 / 270+,
    Snappy 4..11 / 64 / 65, 1-byte / 2-byte offsets) - the speculation's special paths with sharp edges instead of random ones"""
    lls = rng.choice([0, 1, 2, 3, 5, 14, 15, 16, 17, 59, 60, 61, 270, 271, 600], int(rng.integers(1, 5)))
    mls = rng.choice([4, 5, 6, 7, 8, 11, 12, 18, 19, 20, 34, 63, 64, 65, 66, 67, 68, 130, 259, 273, 274, 1000, 70000], int(rng.integers(1, 5)))
    ofs = rng.choice([1, 2, 3, 4, 7, 8, 9, 16, 31, 32, 33, 100, 255, 256, 2047, 2048, 2049, 4096, 20000, 65535, 65536], int(rng.integers(1, 5)))
    out = bytearray(rng.integers(0, 256, int(rng.integers(8, 3000)), dtype=np.uint8).tobytes())
    while len(out) < n:
        out += rng.integers(0, 256, int(rng.choice(lls)), dtype=np.uint8).tobytes()
        d = min(int(rng.choice(ofs)), len(out)); m = int(rng.choice(mls))
        if d >= m: out += out[len(out) - d:len(out) - d + m]
        else:
            seg = bytes(out[-d:]); out += (seg * (m // d + 1))[:m]
    return bytes(out[:n])


def want_lz4(part, acc):
    cap = ol.bound(ol.LZ4, len(part)); w = C.create_string_buffer(max(cap, 1))
    r = L.orc_lz4_compress_fast(part, len(part), w, cap, acc)
    return w.raw[:r]


def want_zstd(part, level):
    cp = lib.ZSTD_getCParams(level, len(part), 0)
    if cp.strategy not in (1, 2): return None
    cap = max(ol.bound(ol.ZSTD, len(part)), 16); w = C.create_string_buffer(cap)
    r = L.orc_zstd_compress_cparams(part, len(part), w, cap, cp.strategy, cp.hashLog, cp.chainLog, cp.minMatch)
    return w.raw[:r]


t_end = time.time() + budget
n_ex = n_chunks = 0
it = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed0 + it); it += 1
    codec = only or ["lz4", "snappy", "zstd"][it % 3]
    size = int(rng.choice([300, 5000, 40000, 65536 + 11, 70000, 140000, 262144, 300000, 400000]))
    raw = seq_shaped(rng, size) if rng.random() < 0.4 else build(rng, size)
    chunk = int(rng.choice([0, 97, 1000, 4096, 65536, 65547, 131072, 262144, int(rng.integers(16, 3000)), int(rng.integers(3000, 70000))]))
    upstream = False
    if codec == "lz4":
        acc = int(rng.choice([1, 1, 1, 2, 5, 64, 70, 3000])); flags = a.api.lz4_acceleration(acc); gc = a.LZ4
        want = lambda p: want_lz4(p, acc)
    elif codec == "snappy":
        flags = int(rng.choice([0, a.api.FLAG_SNAPPY_TABLES_HBM, a.api.FLAG_SNAPPY_TABLES_MIXED])); gc = a.SNAPPY
        want = lambda p: orc.compress(ol.SNAPPY, p)
    else:
        level = int(rng.choice([1, 1, 2, 3, 4])); upstream = bool(rng.integers(0, 2)); gc = a.ZSTD
        lib.aocl_destroy_zstd_encode()
        lib.aocl_setup_zstd_encode(1 if upstream else 0, 0, 0, level, 0)
        flags = a.api.zstd_level(level, upstream)
        if chunk == 0 or chunk > 131072: chunk = 131072
        want = lambda p: want_zstd(p, level)
    tag = f"seed {seed0 + it - 1} {codec} size {size} chunk {chunk} flags {flags:#x}"
    if chunk == 0:
        got = eng.compress_host(gc, raw, flags=flags)
        assert got == want(raw), tag
        assert eng.decompress_host(gc, got, len(raw)) == raw, tag
        n_chunks += 1
    else:
        src = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(eng.device)
        slots, out_len, slot = eng.compress_chunks(gc, src, chunk, flags=flags)
        lens = out_len.cpu().tolist(); host = slots.cpu().numpy()
        ok = True
        for i in range(0, len(raw), chunk):
            k = i // chunk; w = want(raw[i:i + chunk])
            if w is None:
                assert lens[k] == 0, tag; ok = False
                continue
            assert host[k * slot:k * slot + lens[k]].tobytes() == w, tag + f" chunk {k}"
            n_chunks += 1
        if ok:
            dst, dl = eng.decompress_chunks(gc, slots, out_len, slot, len(raw), chunk)
            if not torch.equal(dst, src):                       # keep the input for tools/debug/lz4_dec_diff.py
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                open(os.path.join(ROOT, "gpurun_out", "fuzz_fail.bin"), "wb").write(raw)
            assert torch.equal(dst, src), tag
    n_ex += 1
lib.aocl_destroy_zstd_encode()
lib.aocl_setup_zstd_encode(0, 0, 0, 1, 0)
print(f"fuzz campaign OK: {n_ex} inputs, {n_chunks} chunks compared byte for byte, seeds {seed0}..{seed0 + it - 1}")
