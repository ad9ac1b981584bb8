#!/usr/bin/env python3
"""CPU study for the LZ4 decoder (TEST INFRASTRUCTURE: loads oracle/): how the matches of a block depend on each other when
sequences are executed in groups of G (one sequence per lane).  For every 256 KiB chunk of a sample of the bench corpus the
oracle's LZ4 block is parsed into sequences; per group: matches whose source lies before the group's output ("independent"),
matches that read the group's own output, the exact dependency depth (longest chain of match -> match reads inside the group),
the depth under the conservative rule "wait for every earlier sequence whose output ends behind my source start", literal and
match length classes.  usage: lz4_decode_dep_study.py [MiB] [group sizes ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import aocl_compression_amd as a
import oracle_lib as ol


def parse(block: bytes):
    i, n, seqs = 0, len(block), []
    while i < n:
        t = block[i]; i += 1
        lit = t >> 4
        if lit == 15:
            while True:
                b = block[i]; i += 1; lit += b
                if b != 255: break
        i += lit
        if i >= n: break
        off = block[i] | (block[i + 1] << 8); i += 2
        ml = t & 15
        if ml == 15:
            while True:
                b = block[i]; i += 1; ml += b
                if b != 255: break
        seqs.append((lit, ml + 4, off))
    return seqs


def main():
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    groups = [int(x) for x in sys.argv[2:]] or [8, 32, 64]
    data = a.corpus.silesia_mix(64 << 20)
    orc = ol.Oracle()
    chunk = 262144
    step = max(1, (64 << 20) // chunk // (mib * 4))
    allseq = []
    for c in range(0, (64 << 20) // chunk, step):
        allseq.append(parse(orc.compress(ol.LZ4, data[c * chunk:(c + 1) * chunk].tobytes())))
    nseq = sum(len(s) for s in allseq)
    lits = np.array([q[0] for s in allseq for q in s]); mls = np.array([q[1] for s in allseq for q in s]); offs = np.array([q[2] for s in allseq for q in s])
    print(f"chunks {len(allseq)} sequences {nseq}  mean lit {lits.mean():.2f} mean match {mls.mean():.2f}")
    for name, v, edges in (("literal run", lits, [0, 1, 4, 16, 33, 65, 1 << 30]), ("match length", mls, [4, 16, 33, 65, 257, 1 << 30]), ("offset", offs, [1, 4, 64, 256, 1024, 4096, 16384, 65536])):
        h = np.histogram(v, bins=edges)[0] / len(v)
        print(f"  {name:13s}", "  ".join(f"[{edges[k]},{edges[k+1]}) {h[k]*100:5.1f}%" for k in range(len(h))))
    print(f"  self-overlapping matches (offset < length): {(offs < mls).mean()*100:.1f}%")
    for G in groups:
        dep = tot = 0; depth_exact = []; depth_cons = []
        for s in allseq:
            for g0 in range(0, len(s), G):
                grp = s[g0:g0 + G]
                o = 0; starts = []; ends = []
                for lit, ml, off in grp:
                    starts.append(o + lit); o += lit + ml; ends.append(o)
                lev_e = [0] * len(grp); lev_c = [0] * len(grp)
                for i, (lit, ml, off) in enumerate(grp):
                    S = starts[i] - off; E = min(S + ml, starts[i])
                    tot += 1
                    if E <= 0: continue
                    dep += 1
                    le = lc = 1
                    for j in range(i):
                        if ends[j] > S: lc = max(lc, lev_c[j] + 1)                      # conservative: every earlier sequence that ends behind S
                        if starts[j] < E and ends[j] > S: le = max(le, lev_e[j] + 1)     # exact: the match output of j is read
                    lev_e[i] = le; lev_c[i] = lc
                depth_exact.append(max(lev_e)); depth_cons.append(max(lev_c))
        de = np.array(depth_exact); dc = np.array(depth_cons)
        print(f"group {G:3d}: matches reading the group's own output {dep/tot*100:5.1f}%  (= {dep/len(de):.1f} per group)   depth exact mean {de.mean():.2f} p90 {np.percentile(de,90):.0f} max {de.max()}   conservative mean {dc.mean():.2f} p90 {np.percentile(dc,90):.0f} max {dc.max()}")


if __name__ == "__main__":
    main()
