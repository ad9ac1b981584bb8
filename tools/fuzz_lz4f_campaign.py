#!/usr/bin/env python3
"""Random-seed campaign for the LZ4 frame API (TEST INFRASTRUCTURE: loads oracle/_ref): LZ4F_compressFrame of the GPU library
against the reference's with random preferences (block size, linked / independent blocks, all checksum flags, content size,
levels -9 .. 2) on random inputs - byte for byte - and both libraries' frames decoded by the other one.
usage: tools/fuzz_lz4f_campaign.py <seconds> [seed]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import aocl_compression_amd as a
import oracle_lib as ol
import lz4f_lib as lf
import zstd_frame_cases as zfc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
eng = a.engine(0); G = lf.bind(eng.lib); R = lf.bind(C.CDLL(ol.REF_SO))
src_code = open(os.path.join(ROOT, "tools", "fuzz_campaign.py")).read()
ns = {"np": np}
exec(src_code[src_code.index("def build(rng, target):"):src_code.index("def want_lz4")], ns)
t0 = time.time(); it = 0; linked = 0
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
print(f"LZ4F campaign OK: {it} frames ({linked} with linked blocks), seeds {seed0}..{seed0 + it - 1}")
