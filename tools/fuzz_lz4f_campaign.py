#!/usr/bin/env python3
"""Random-seed campaign for the LZ4 frame API (TEST INFRASTRUCTURE: loads oracle/_ref): LZ4F_compressFrame of the GPU library
against the reference's with random preferences (block size, linked / independent blocks, all checksum flags, content size,
levels -9 .. 2) on random inputs - byte for byte - and both libraries' frames decoded by the other one; and the same input
through LZ4F_compressBegin / Update / flush / End on one cctx per library (random pieces handed over in place or as copies,
random flushes, autoFlush, stableSrc, two frames per cctx): the streams of linked blocks must equal the reference's too.
usage: tools/fuzz_lz4f_campaign.py <seconds> [seed]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import aocl_compression_amd as a
import oracle_lib as ol
import lz4f_lib as lf
import zstd_frame_cases as zfc
import test_lz4frame_stream_linked as tsl

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
eng = a.engine(0); G = lf.bind(eng.lib); R = lf.bind(C.CDLL(ol.REF_SO))
src_code = open(os.path.join(ROOT, "tools", "fuzz_campaign.py")).read()
ns = {"np": np}
exec(src_code[src_code.index("def build(rng, target):"):src_code.index("def want_lz4")], ns)
t0 = time.time(); it = 0; linked = 0; streams = 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed0 + it); it += 1
    n = int(rng.choice([rng.integers(0, 3000), rng.integers(3000, 70000), rng.integers(65537, 700000), rng.integers(700000, 5000000)]))
    k = int(rng.integers(0, 4))
    if n == 0: data = b""
    elif k == 0: data = ns["build"](rng, n)
    elif k == 1: data = ns["seq_shaped"](rng, n)
    elif k == 2: data = zfc.make("mixed", n, int(rng.integers(0, 1 << 30)))
    else: data = a.corpus.make(list(a.corpus.KINDS)[int(rng.integers(0, len(a.corpus.KINDS)))], n, seed=int(rng.integers(0, 1 << 30))).tobytes()
    p = lf.prefs(bsid=int(rng.choice([0, 4, 5, 6, 7])), independent=bool(rng.integers(0, 2)), content_checksum=bool(rng.integers(0, 2)),
                 block_checksum=bool(rng.integers(0, 2)), content_size=bool(rng.integers(0, 2)), level=int(rng.choice([0, 0, 1, 2, -1, -4, -9])))
    got, want = lf.compress(G, data, p), lf.compress(R, data, p)
    tag = f"seed {seed0 + it - 1} n {n} bsid {p.frameInfo.blockSizeID} mode {p.frameInfo.blockMode} level {p.compressionLevel}"
    assert isinstance(got, bytes) and got == want, tag + ": frame differs from the reference's"
    assert lf.decompress(G, want, n) == (data, len(want)), tag + ": GPU decode of the reference's frame"
    if not p.frameInfo.blockMode and len(got) > 7 and not (got[4] >> 5) & 1: linked += 1
    if n >= 2 and n <= 2500000:                               # the streaming calls, two frames on one cctx
        buf = (C.c_char * n).from_buffer_copy(data)
        jobs = []
        for f in range(2):
            total = int(rng.integers(1, n + 1)); start = int(rng.integers(0, n - total + 1))
            lo, hi = [(1, 3000), (1, 150000), (60000, 70000), (1, 400000)][int(rng.integers(0, 4))]
            q = lf.prefs(bsid=int(rng.choice([0, 4, 4, 5, 6])), independent=bool(rng.integers(0, 4) == 0), content_checksum=bool(rng.integers(0, 2)),
                         block_checksum=bool(rng.integers(0, 2)), level=int(rng.choice([0, 0, 1, -1, -6])), auto_flush=bool(rng.integers(0, 2)))
            pieces = tsl.cut(rng, total, lo, hi)
            flush_at = set(int(x) for x in rng.integers(0, len(pieces), 2)) if rng.integers(0, 3) == 0 else set()
            stable = bool(rng.integers(0, 6) == 0)
            jobs.append((q, start, pieces, flush_at, stable, bool(rng.integers(0, 2)) and not stable))
        want2, got2 = tsl.run(R, buf, jobs), tsl.run(G, buf, jobs)
        assert got2 == want2, tag + f": streamed frames differ ({[(j[1], len(j[2]), sorted(j[3]), j[4], j[5], j[0].frameInfo.blockMode, j[0].autoFlush, j[0].frameInfo.blockSizeID) for j in jobs]})"
        streams += 2
print(f"LZ4F campaign OK: {it} frames ({linked} with linked blocks), {streams} streamed frames, seeds {seed0}..{seed0 + it - 1}")
