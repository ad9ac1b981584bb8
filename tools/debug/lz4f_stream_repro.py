"""Replays one seed of tools/fuzz_lz4f_campaign.py (the streamed frames) and says which frame and block differ from the reference.
usage (GPU box): python tools/debug/lz4f_stream_repro.py <seed>"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import aocl_compression_amd as a
import oracle_lib as ol
import lz4f_lib as lf
import zstd_frame_cases as zfc
import test_lz4frame_stream_linked as tsl
seed = int(sys.argv[1])
eng = a.engine(0); G = lf.bind(eng.lib); R = lf.bind(C.CDLL(ol.REF_SO))
src_code = open(os.path.join(ROOT, "tools", "fuzz_campaign.py")).read()
ns = {"np": np}
exec(src_code[src_code.index("def build(rng, target):"):src_code.index("def want_lz4")], ns)
rng = np.random.default_rng(seed)
n = int(rng.choice([rng.integers(0, 3000), rng.integers(3000, 70000), rng.integers(65537, 700000), rng.integers(700000, 5000000)]))
k = int(rng.integers(0, 4))
if n == 0: data = b""
elif k == 0: data = ns["build"](rng, n)
elif k == 1: data = ns["seq_shaped"](rng, n)
elif k == 2: data = zfc.make("mixed", n, int(rng.integers(0, 1 << 30)))
else: data = a.corpus.make(list(a.corpus.KINDS)[int(rng.integers(0, len(a.corpus.KINDS)))], n, seed=int(rng.integers(0, 1 << 30))).tobytes()
p = lf.prefs(bsid=int(rng.choice([0, 4, 5, 6, 7])), independent=bool(rng.integers(0, 2)), content_checksum=bool(rng.integers(0, 2)),
             block_checksum=bool(rng.integers(0, 2)), content_size=bool(rng.integers(0, 2)), level=int(rng.choice([0, 0, 1, 2, -1, -4, -9])))
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
    print("frame", f, "start", start, "total", total, "pieces", pieces[:12], "flush", sorted(flush_at), "stable", stable, "copies", jobs[-1][5], "mode", q.frameInfo.blockMode, "bsid", q.frameInfo.blockSizeID, "level", q.compressionLevel, "autoFlush", q.autoFlush, "bcrc", q.frameInfo.blockChecksumFlag)
want2, got2 = tsl.run(R, buf, jobs), tsl.run(G, buf, jobs)
for f in range(2):
    w, g = want2[f], got2[f]
    print("frame", f, "equal", w == g, len(w), len(g))
    if w != g:
        d = next((j for j in range(min(len(g), len(w))) if g[j] != w[j]), -1)
        print("  first diff at", d, g[d:d+12].hex(), w[d:d+12].hex())
        # walk the blocks of the reference frame
        pos = 7
        bi = 0
        while pos + 4 <= len(w):
            bsz = int.from_bytes(w[pos:pos+4], "little")
            if bsz == 0: break
            ln = bsz & 0x7FFFFFFF
            if pos <= d < pos + 4 + ln + (4 if jobs[f][0].frameInfo.blockChecksumFlag else 0):
                print("  in block", bi, "at frame offset", pos, "block size", ln, "stored" if bsz >> 31 else "")
            pos += 4 + ln + (4 if jobs[f][0].frameInfo.blockChecksumFlag else 0); bi += 1
# each frame alone on a fresh cctx
for f in range(2):
    w1, g1 = tsl.run(R, buf, [jobs[f]])[0], tsl.run(G, buf, [jobs[f]])[0]
    print("frame", f, "alone: equal", w1 == g1, "ref alone == ref in sequence", w1 == want2[f])
