#!/usr/bin/env python3
"""Debug helper: one input of tools/fuzz_lz4f_campaign.py by seed; where the GPU frame and the reference's frame part."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import aocl_compression_amd as a
import oracle_lib as ol
import lz4f_lib as lf
import zstd_frame_cases as zfc
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
got, want = lf.compress(G, data, p), lf.compress(R, data, p)
print("n", n, "kind", k, "bsid", p.frameInfo.blockSizeID, "mode", p.frameInfo.blockMode, "level", p.compressionLevel, "sizes", len(got), len(want), "equal", got == want)
if got != want:
    d = next(i for i in range(min(len(got), len(want))) if got[i] != want[i])
    # walk the reference frame's blocks to name the block
    flg = want[4]; pos = 7 + (8 if (flg >> 3) & 1 else 0) + (4 if flg & 1 else 0)
    blk = 0; bsz = {4: 65536, 5: 262144, 6: 1 << 20, 7: 4 << 20}.get(p.frameInfo.blockSizeID or 4)
    while pos + 4 <= len(want):
        h = int.from_bytes(want[pos:pos + 4], "little"); sz = h & 0x7FFFFFFF
        if sz == 0: break
        if pos <= d < pos + 4 + sz + (4 if (flg >> 4) & 1 else 0):
            print("first difference at frame byte", d, "= block", blk, "(source offset", blk * bsz, ") byte", d - pos - 4, "of", sz, "raw" if h >> 31 else "compressed")
            print("got ", got[max(pos + 4, d - 16):d + 16].hex()); print("want", want[max(pos + 4, d - 16):d + 16].hex())
            break
        pos += 4 + sz + (4 if (flg >> 4) & 1 else 0); blk += 1
    def seqs(blob):
        out = []; i = 0; n_ = len(blob); o = 0
        while i < n_:
            t = blob[i]; i += 1; lit = t >> 4
            if lit == 15:
                while True:
                    b = blob[i]; i += 1; lit += b
                    if b != 255: break
            i += lit; o += lit
            if i >= n_: out.append((o - lit, lit, 0, 0)); break
            off = blob[i] | (blob[i + 1] << 8); i += 2; ml = t & 15
            if ml == 15:
                while True:
                    b = blob[i]; i += 1; ml += b
                    if b != 255: break
            ml += 4; out.append((o - lit, lit, ml, off)); o += ml
        return out
    def block(fr, idx):
        flg = fr[4]; pos = 7 + (8 if (flg >> 3) & 1 else 0) + (4 if flg & 1 else 0)
        for b in range(idx + 1):
            h = int.from_bytes(fr[pos:pos + 4], "little"); sz = h & 0x7FFFFFFF
            if b == idx: return fr[pos + 4:pos + 4 + sz], h >> 31
            pos += 4 + sz + (4 if (flg >> 4) & 1 else 0)
    gb, graw = block(got, blk); wb, wraw = block(want, blk)
    if not graw and not wraw:
        gs, ws = seqs(gb), seqs(wb)
        k = next((i for i in range(min(len(gs), len(ws))) if gs[i] != ws[i]), min(len(gs), len(ws)))
        print("sequences", len(gs), len(ws), "first differing sequence", k)
        for i in range(max(0, k - 2), min(k + 3, max(len(gs), len(ws)))):
            print(i, "got", gs[i] if i < len(gs) else None, "want", ws[i] if i < len(ws) else None)
