#!/usr/bin/env python3
"""Random-seed parity campaign aimed at the Zstd entropy stage on the GPU box (TEST INFRASTRUCTURE: loads oracle/): inputs are
shuffled bytes drawn to a prescribed histogram, so nearly everything is a literal and the Huffman builder sees count vectors
full of ties, counts around the sort's cutoff (164 / 165 / 166) and around powers of two, tiny and full alphabets, sections
that need the height limiter.  Batches of equal-size chunks through aoclgpu_zstd_compress_chunks (levels 1 and 3) against the
oracle, byte for byte.  usage: tools/fuzz_entropy_campaign.py <seconds> [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import aocl_compression_amd as a
import oracle_lib as ol

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
eng = a.engine(0); orc = ol.Oracle()
import ctypes as C


class CP(C.Structure):
    _fields_ = [(k, C.c_uint) for k in ("windowLog", "chainLog", "hashLog", "searchLog", "minMatch", "targetLength")] + [("strategy", C.c_int)]


eng.lib.ZSTD_getCParams.restype = CP
eng.lib.ZSTD_getCParams.argtypes = [C.c_int, C.c_ulonglong, C.c_size_t]
orc.L.orc_zstd_compress_cparams.restype = C.c_size_t
orc.L.orc_zstd_compress_cparams.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int, C.c_uint, C.c_uint, C.c_uint]


def counts_for(rng, n):
    k = int(rng.integers(0, 6))
    nsym = int(rng.choice([2, 3, 5, 16, 40, 100, 200, 256]))
    if k == 0:      # around the cutoff of the reference's bucket sort
        c = rng.integers(160, 170, nsym)
    elif k == 1:    # a handful of distinct values: many ties
        vals = rng.choice([1, 2, 3, 4, 7, 8, 9, 15, 16, 17, 63, 64, 65, 127, 128, 129, 163, 164, 165, 166, 255, 256, 257, 511, 512, 1023, 1024, 5000],
                          int(rng.integers(1, 6)))
        c = rng.choice(vals, nsym)
    elif k == 2:    # power law: long tail of rare symbols (height limiter)
        c = np.maximum(1, (n * (np.arange(1, nsym + 1) ** -rng.uniform(1.0, 3.0))).astype(np.int64))
    elif k == 3:    # geometric
        c = np.maximum(1, (n * 0.5 ** (np.arange(nsym) / rng.uniform(0.5, 8))).astype(np.int64))
    elif k == 4:    # near-flat with noise inside one log2 bucket
        base = int(rng.choice([130, 200, 256, 300, 511, 600]))
        c = base + rng.integers(0, 12, nsym)
    else:           # a few heavy symbols + ties at 164
        c = np.r_[rng.integers(2000, 20000, min(nsym, 4)), np.full(max(nsym - 4, 0), 164)]
    c = np.asarray(c, dtype=np.int64)
    syms = rng.choice(256, len(c), replace=False)
    full = np.zeros(256, dtype=np.int64); full[syms] = c
    tot = int(full.sum())
    if tot > n:                                   # scale down, keep the shape
        full = np.maximum((full * n) // tot, (full > 0).astype(np.int64))
    return full


def seq_shaped(rng, n):
    """a chunk whose parse is (mostly) prescribed: fresh random literals of a few fixed run lengths, each followed by a copy of a
    few fixed lengths at a few fixed distances - sequence-code histograms with one to four symbols per stream (RLE / predefined /
    compressed tables with ties, the `mostFrequent` thresholds of ZSTD_selectEncodingType)"""
    lls = rng.choice([0, 1, 2, 3, 5, 8, 15, 16, 17, 24, 40, 64, 130, 600], int(rng.integers(1, 5)))
    mls = rng.choice([5, 6, 7, 8, 9, 12, 18, 34, 35, 36, 40, 66, 67, 130, 131, 259, 1000], int(rng.integers(1, 5)))
    ofs = rng.choice([1, 2, 3, 4, 7, 8, 9, 16, 31, 32, 33, 100, 255, 256, 1000, 4096, 20000, 65536], int(rng.integers(1, 5)))
    out = bytearray(rng.integers(0, 256, int(rng.integers(8, 3000)), dtype=np.uint8).tobytes())
    while len(out) < n:
        out += rng.integers(0, 256, int(rng.choice(lls)), dtype=np.uint8).tobytes()
        d = min(int(rng.choice(ofs)), len(out)); m = int(rng.choice(mls))
        if d >= m: out += out[len(out) - d:len(out) - d + m]
        else:
            seg = bytes(out[-d:]); out += (seg * (m // d + 1))[:m]
    return bytes(out[:n])


# the level table ZSTD_getCParams answers from must be the one the batch flags select (AOCL's): as in the reference, the library
# starts with the upstream table until the dispatcher has run (csrc/zstd_api.cpp)
eng.lib.aocl_setup_zstd_encode.restype = C.c_void_p
eng.lib.aocl_setup_zstd_encode.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t]
eng.lib.aocl_destroy_zstd_encode()
eng.lib.aocl_setup_zstd_encode(0, 0, 0, 3, 0)
t0 = time.time(); it = 0; chunks = 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed0 + it); it += 1
    # sizes: a few bulk ones and the neighbourhoods of the literal-section thresholds (raw below 64, one stream below 256,
    # header formats at 1 KiB and 16 KiB, table reuse preference at 1 KiB, sampling of suspected-incompressible input from
    # 40 960, 4-stream segments of 64 KiB)
    near = int(rng.choice([64, 256, 1024, 4096, 16384, 40960, 65536, 65792, 131072]))
    n_target = int(rng.choice([2000, 20000, 50000, 131072, max(8, min(131072, near + int(rng.integers(-12, 13))))]))
    parts = []
    for _ in range(32):
        if rng.random() < 0.35:
            parts.append(seq_shaped(rng, n_target)); continue
        full = counts_for(rng, n_target)
        data = rng.permutation(np.repeat(np.arange(256, dtype=np.uint8), full))
        if len(data) < n_target:                  # top up with copies of the most frequent symbol: counts stay tied elsewhere
            data = np.r_[data, np.full(n_target - len(data), int(np.argmax(full)), dtype=np.uint8)]
            data = rng.permutation(data)
        parts.append(data[:n_target].tobytes())
    blob = b"".join(parts)
    src = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(eng.device)
    for level in (1, 3):
        slots, lens, slot = eng.compress_chunks(a.ZSTD, src, n_target, flags=a.api.zstd_level(level))
        host = slots.cpu().numpy(); ln = lens.cpu().numpy()
        for i, p in enumerate(parts):
            got = host[i * slot:i * slot + int(ln[i])].tobytes()
            if level == 1:
                want = orc.compress(ol.ZSTD, p, level=1)
            else:
                cp = eng.lib.ZSTD_getCParams(level, len(p), 0)
                cap = ol.bound(ol.ZSTD, len(p)); w = C.create_string_buffer(max(cap, 16))
                r = orc.L.orc_zstd_compress_cparams(p, len(p), w, max(cap, 16), 2, cp.hashLog, cp.chainLog, cp.minMatch)
                want = w.raw[:r]
            if got != want:
                print("MISMATCH seed", seed0 + it - 1, "chunk", i, "n", n_target, "level", level, len(got), len(want)); sys.exit(1)
            chunks += 1
print(f"OK seed0={seed0} batches={it} chunks={chunks} in {time.time() - t0:.0f}s")
