#!/usr/bin/env python3
"""Development aid: the first sequence in which the GPU's LZ4 block differs from the oracle's, with its input position.
usage: tools/debug/lz4_seq_diff.py <kind> <size> [seed] [offset into a 2 MiB + 777 corpus]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import aocl_compression_amd as a
import oracle_lib as ol


def seqs(comp):
    n = len(comp); p = 0; pos = 0; out = []
    while p < n:
        t = comp[p]; p += 1; lit = t >> 4
        if lit == 15:
            while True:
                e = comp[p]; p += 1; lit += e
                if e != 255: break
        p += lit
        if p >= n: out.append((pos, lit, 0, 0)); break
        off = comp[p] | (comp[p + 1] << 8); p += 2; ml = t & 15
        if ml == 15:
            while True:
                e = comp[p]; p += 1; ml += e
                if e != 255: break
        out.append((pos, lit, ml + 4, off)); pos += lit + ml + 4
    return out


kind, size = sys.argv[1], int(sys.argv[2])
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 11
eng = a.engine(0); orc = ol.Oracle()
data = a.corpus.make(kind, (2 << 20) + 777, seed=seed).tobytes()
for c in range(0, len(data), size):
    raw = data[c:c + size]
    g = eng.compress_host(a.LZ4, raw); w = orc.compress(ol.LZ4, raw)
    if g == w: continue
    sg, sw = seqs(g), seqs(w)
    for i, (x, y) in enumerate(zip(sg, sw)):
        if x != y:
            print(f"chunk at {c}: sequence {i}: gpu (pos, lit, ml, off) = {x}  oracle = {y}  | before: {sw[max(0, i - 2):i]}")
            break
    break
else:
    print("all chunks equal")
