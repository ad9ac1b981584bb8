#!/usr/bin/env python3
"""CPU model (round 5) of a SPECULATIVE SEGMENTED token scan for the LZ4 decoder: the 64 lanes of a wave each parse the token
chain of their own G-byte segment of the compressed input sequentially, starting WARM bytes in front of it at an arbitrary byte
(a token parser that starts at a wrong byte falls into step with the real chain after a few tokens); lane k is right when the
position its predecessor leaves segment k-1 at is one of the token starts lane k visited, lanes that are not run again from
that position, until nothing changes.  Reports, per corpus kind: iterations of the longest lane (what the wave pays), lanes that
needed a repair, repair rounds, tokens per round - against the sequential chain (exactness is asserted).
Usage: lz4_spec_scan_model.py [G] [WARM]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import aocl_compression_amd as a
import oracle_lib as ol

G = int(sys.argv[1]) if len(sys.argv) > 1 else 32
WARM = int(sys.argv[2]) if len(sys.argv) > 2 else 128
orc = ol.Oracle()


def parse(buf, p, n):
    """token at p -> (next token start, hard); hard: a length field of more than one extension byte, or input too short"""
    if p + 1 >= n: return p, True
    t = buf[p]; lit = t >> 4; q = p + 1
    if lit == 15:
        e = buf[q]; q += 1
        if e > 240: return p, True
        lit += e
    q += lit
    if q + 3 >= n: return p, True
    ml = t & 15; q2 = q + 2
    if ml == 15:
        e = buf[q2]; q2 += 1
        if e > 240: return p, True
    return q2, False


def true_chain(buf, n):
    s = []; p = 0
    while True:
        nx, hard = parse(buf, p, n)
        if hard: break
        s.append(p); p = nx
    return s, p


def model(buf, n, stats):
    chain, stop = true_chain(buf, n)
    chain_set = set(chain)
    ip = 0
    while ip < stop:
        segs = [(ip + G * k, ip + G * (k + 1)) for k in range(64)]
        visited = [set() for _ in range(64)]; exit_ = [0] * 64; iters = [0] * 64
        def run(k, start):
            p = start; v = set(); it = 0
            lo, hi = segs[k]
            while p < hi:
                nx, hard = parse(buf, p, n)
                it += 1
                if hard: break
                if p >= lo: v.add(p)
                p = nx
            visited[k] = v; exit_[k] = p; iters[k] = it
        for k in range(64):
            run(k, max(ip, segs[k][0] - WARM))
        stats["iters"].append(max(iters)); stats["rounds"] += 1
        reps = 0; repaired = set()
        while True:
            bad = []
            for k in range(64):
                inn = ip if k == 0 else exit_[k - 1]
                lo, hi = segs[k]
                if inn >= hi:                      # the predecessor jumped over the whole segment (or stopped): pass through
                    if exit_[k] != inn or visited[k]: visited[k] = set(); exit_[k] = inn
                    continue
                if inn < lo: inn = lo if False else inn   # (cannot happen: exits are >= the segment's end)
                if inn in visited[k] or (inn == exit_[k] and not [x for x in visited[k] if x >= inn]): continue
                bad.append((k, inn))
            if not bad: break
            reps += 1
            mx = 0
            for k, inn in bad[:]:                  # every wrong lane runs again from where its predecessor left off (all at once)
                run(k, inn); repaired.add(k); mx = max(mx, iters[k])
            stats["rep_iters"].append(mx)
        stats["repairs"].append(reps); stats["bad_lanes"].append(len(repaired))
        # the round's tokens: per lane those at or behind its entry
        toks = []
        for k in range(64):
            inn = ip if k == 0 else exit_[k - 1]
            toks += sorted(x for x in visited[k] if x >= inn)
        want = [c for c in chain if ip <= c < max(ip + 64 * G, ip + 1) and c < exit_[63]] if True else None
        end = exit_[63]
        want = [c for c in chain if ip <= c < end]
        assert toks == want, (ip, len(toks), len(want))
        stats["tokens"].append(len(toks))
        if end == ip: break
        ip = end
        if ip not in chain_set: break              # the chain stopped at a hard token: the sequential path takes over (model: end)


for kind in ["text", "jsonlog", "binary", "database"]:
    data = a.corpus.make(kind, 1 << 20, seed=7).tobytes()
    stats = {"iters": [], "repairs": [], "bad_lanes": [], "tokens": [], "rep_iters": [], "rounds": 0}
    for c in range(0, len(data), 262144):
        comp = orc.compress(ol.LZ4, data[c:c + 262144])
        model(comp, len(comp), stats)
    it, rp, bl, tk = (np.array(stats[x]) for x in ("iters", "repairs", "bad_lanes", "tokens"))
    ri = np.array(stats["rep_iters"]) if stats["rep_iters"] else np.array([0])
    print(f"{kind:9s} G={G} WARM={WARM}: rounds {stats['rounds']:5d}  tokens/round {tk.mean():6.1f}  longest lane {it.mean():5.1f} iterations (max {it.max()})  "
          f"lanes repaired/round {bl.mean():5.2f}  repair rounds {rp.mean():4.2f} (max {rp.max()}), {ri.mean():4.1f} iterations each  "
          f"-> {(it.sum() + ri.sum()) / tk.sum():.3f} lane-iterations of the wave per token")
