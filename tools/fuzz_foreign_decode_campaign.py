#!/usr/bin/env python3
"""Random-seed campaign for the DECODERS on streams this library did not write (TEST INFRASTRUCTURE: loads oracle/_ref, the
reference built in place): Zstd frames from the reference's ZSTD_compress at every level (-7 .. 22: all match finders,
repeated FSE tables, treeless literals, 3-byte matches, multi-block frames with windows up to 8 MiB, frames with checksums
through ZSTD_compress_advanced) and LZ4 blocks from LZ4_compress_HC (levels 3 - 12: optimal parser) and LZ4_compress_fast,
decoded on the GPU and compared with the source.  usage: tools/fuzz_foreign_decode_campaign.py <seconds> [seed]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import aocl_compression_amd as a
import oracle_lib as ol
import zstd_frame_cases as zfc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
eng = a.engine(0); ref = ol.Ref(); R = ref.L; lib = eng.lib
src_code = open(os.path.join(ROOT, "tools", "fuzz_frames_campaign.py")).read()
ns = {"np": np, "zfc": zfc}
exec(src_code[src_code.index("def grammar"):src_code.index("def make_input")], ns)       # the LZ-grammar generator
grammar = ns["grammar"]
R.ZSTD_compress.restype = C.c_size_t
R.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
R.ZSTD_isError.restype = C.c_uint
R.ZSTD_isError.argtypes = [C.c_size_t]
R.LZ4_compress_HC.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int]
R.LZ4_compress_fast.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int]

t0 = time.time(); it = 0; nz = nl = 0; nbytes = 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed0 + it); it += 1
    n = int(rng.choice([rng.integers(0, 300), rng.integers(300, 70000), rng.integers(70000, 600000), rng.integers(600000, 5000000)]))
    kind = int(rng.integers(0, 3))
    if kind == 0: data = grammar(rng, n) if n else b""
    elif kind == 1: data = zfc.make("mixed", n, int(rng.integers(0, 1 << 30))) if n else b""
    else: data = a.corpus.make(list(a.corpus.KINDS)[int(rng.integers(0, len(a.corpus.KINDS)))], n, seed=int(rng.integers(0, 1 << 30))).tobytes() if n else b""
    n = len(data)
    # ---- Zstd: the reference's frame at a random level ----
    big = n > 1500000
    level = int(rng.choice([-7, -3, -1, 1, 2, 3, 4, 5, 6, 7, 9, 12] if big else [-7, -3, -1, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 16, 17, 18, 19, 20, 22]))
    if rng.random() < 0.25:
        cp = zfc.ref_level_cparams(ref, level if level > 0 else 1, max(n, 1))
        cp.windowLog = int(rng.integers(10, 24))
        if cp.strategy >= 6: cp.chainLog = min(cp.chainLog, cp.windowLog + 1)
        try:
            frame = zfc.ref_frame(ref, data, cp)                   # explicit window: window descriptor, small blocks
        except AssertionError:
            frame = None
    else:
        cap = n + (n >> 7) + 1024; ob = C.create_string_buffer(cap)
        r = R.ZSTD_compress(ob, cap, data, n, level)
        frame = None if R.ZSTD_isError(r) else ob.raw[:r]
    if frame is not None:
        back = C.create_string_buffer(max(n, 1))
        rr = lib.aoclgpu_zstd_decompress_host_frames(frame, len(frame), back, n)
        if rr != n or back.raw[:n] != data:
            print("ZSTD FOREIGN FRAME MISMATCH seed", seed0 + it - 1, "n", n, "level", level, "kind", kind, rr); sys.exit(1)
        nz += 1; nbytes += n
    # ---- LZ4: the reference's HC / fast blocks ----
    if 0 < n <= 4000000:
        cap = ol.bound(ol.LZ4, n); ob = C.create_string_buffer(cap)
        if rng.random() < 0.7: r = R.LZ4_compress_HC(data, ob, n, cap, int(rng.integers(3, 13)))
        else: r = R.LZ4_compress_fast(data, ob, n, cap, int(rng.choice([1, 2, 9, 100])))
        assert r > 0
        got = eng.decompress_host(a.LZ4, ob.raw[:r], n)
        if got != data:
            print("LZ4 FOREIGN BLOCK MISMATCH seed", seed0 + it - 1, "n", n); sys.exit(1)
        nl += 1
print(f"OK seed0={seed0} inputs={it} zstd_frames={nz} lz4_blocks={nl} zstd_bytes={nbytes} in {time.time() - t0:.0f}s")
