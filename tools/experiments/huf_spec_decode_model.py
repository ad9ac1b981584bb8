#!/usr/bin/env python3
"""CPU model of a speculative 16-lanes-per-stream Huffman literal decoder for the Zstd decoder (groundwork for round 3; see
huf_selfsync_study.py for the statistics that motivate it).  TEST INFRASTRUCTURE: loads oracle/.

Scheme per stream (a backward bitstream with a start mark, read from bit `top` down to bit 0):
  1. cut [0, top] into LANES segments; lane j guesses that a code starts at its segment's upper end e_j (lane 0 knows: e_0 = top);
  2. every lane decodes from its start until it leaves its segment, and records where: x_j = first code start <= e_{j+1};
  3. lane j+1 adopts x_j as its start when it differs from the start it used; repeat 2-3 for the lanes whose start changed,
     until no start changes (a misplaced decoder falls into step after a few codes, so x_j usually does not depend on the guess
     and one repair round settles everything);
  4. prefix sum of the lanes' symbol counts = where each lane writes; every lane decodes its segment once more and writes.
The model runs this scheme literally and checks the symbols against the plain sequential decode, on the literal streams the
level-1 compressor writes for the bench corpus kinds, and reports how many rounds step 3 needed and the total decode work
relative to one sequential pass.  usage: python tools/experiments/huf_spec_decode_model.py [chunks per kind]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import aocl_compression_amd as a
import oracle_lib as ol
from huf_selfsync_study import literal_streams, orc

LANES = 16


def tables(lens, tl):
    nb = np.zeros(1 << tl, dtype=np.int64); sym = np.zeros(1 << tl, dtype=np.int64)
    pos = 0
    for w in range(1, tl + 1):
        for s in range(256):
            if lens[s] and tl + 1 - lens[s] == w:
                nb[pos:pos + (1 << (w - 1))] = lens[s]; sym[pos:pos + (1 << (w - 1))] = s; pos += 1 << (w - 1)
    return nb, sym


class Stream:
    def __init__(self, raw, tl):
        bits = np.unpackbits(np.frombuffer(raw, dtype=np.uint8), bitorder="little").astype(np.int64)
        top = len(bits) - 1
        while not bits[top]: top -= 1
        self.top, self.tl = top, tl
        # value of the tl bits just below position p, for every p (zeros below bit 0), MSB = the bit right below p
        pad = np.concatenate([np.zeros(tl, dtype=np.int64), bits[:top]])
        self.code = np.zeros(top + 1, dtype=np.int64)
        for k in range(tl):
            self.code[:] += pad[tl - 1 - k: tl - 1 - k + top + 1] << (tl - 1 - k) if False else 0
        # (straightforward, slower form)
        w = 1 << np.arange(tl - 1, -1, -1)
        for p in range(top + 1):
            self.code[p] = int(pad[p:p + tl][::-1].dot(w))


def sequential(st, nb, sym):
    out = []; p = st.top
    while p > 0:
        c = st.code[p]; out.append(int(sym[c])); p -= int(nb[c])
    return out, p


def speculative(st, nb, sym, lanes=LANES):
    top = st.top
    seg = -(-top // lanes)
    start = [max(top - j * seg, 0) for j in range(lanes)]        # e_j; lane 0 is exact
    end_limit = [max(top - (j + 1) * seg, 0) for j in range(lanes)]
    work = 0; rounds = 0
    exitp = [None] * lanes
    dirty = list(range(lanes))
    while dirty:
        rounds += 1
        nxt = []
        for j in dirty:
            p = start[j]
            while p > end_limit[j]:
                p -= int(nb[st.code[p]]); work += 1
            exitp[j] = p
        for j in dirty:
            if j + 1 < lanes and exitp[j] != start[j + 1] and start[j + 1] > 0:
                start[j + 1] = exitp[j]; nxt.append(j + 1)
            elif j + 1 < lanes and start[j + 1] == 0 and exitp[j] > 0:
                start[j + 1] = exitp[j]; nxt.append(j + 1)
        dirty = sorted(set(nxt))
    out = []
    for j in range(lanes):                                        # step 4 (the writes; counted as one more pass)
        p = start[j]
        while p > end_limit[j] and p > 0:
            c = st.code[p]; out.append(int(sym[c])); p -= int(nb[c]); work += 1
    return out, rounds, work


if __name__ == "__main__":
    per_kind = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    print("kind            streams  max repair rounds  decode steps / sequential steps")
    for kind in sorted(a.corpus.KINDS):
        data = a.corpus.make(kind, per_kind * 131072, seed=5).tobytes()
        nst = 0; mr = 0; w_spec = 0; w_seq = 0
        for i in range(per_kind):
            part = data[i * 131072:(i + 1) * 131072]
            ls = literal_streams(orc.compress(ol.ZSTD, part), len(part))
            if ls is None: continue
            lens, tl, streams = ls
            nb, sym = tables(lens, tl)
            for raw in streams:
                if len(raw) < 64: continue
                st = Stream(raw, tl)
                want, endp = sequential(st, nb, sym)
                got, rounds, work = speculative(st, nb, sym)
                assert got == want, (kind, i, len(raw))
                nst += 1; mr = max(mr, rounds); w_spec += work; w_seq += len(want)
        if nst:
            print(f"{kind:14s} {nst:8d} {mr:18d} {w_spec / w_seq:12.2f}   (per lane: {w_spec / w_seq / LANES:.2f} of a sequential pass)")
