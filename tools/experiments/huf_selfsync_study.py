#!/usr/bin/env python3
"""CPU study for a speculative (self-synchronising) Huffman literal decoder - groundwork for the Zstd decoder's 30 % literal
stage.  TEST INFRASTRUCTURE: loads oracle/.

A Zstd Huffman stream is read backwards from a start mark; a decoder that starts at an arbitrary bit offset produces garbage
until it happens to land on a true code boundary, and from then on it is in step with the real decoder.  If that happens
quickly, a stream can be cut into 16 segments decoded by 16 lanes at once (each lane keeps what follows its first agreement
with its predecessor).  This script measures, on the literal streams the level-1 compressor actually writes for the bench
corpus, how many bits / symbols a misplaced decoder needs to fall into step.

usage: python tools/experiments/huf_selfsync_study.py [chunks per kind]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import aocl_compression_amd as a
import oracle_lib as ol

orc = ol.Oracle()
orc.L.orc_dbg_huf_lengths.restype = C.c_uint
orc.L.orc_dbg_huf_lengths.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
per_kind = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rng = np.random.default_rng(1)


def literal_streams(frame, n):
    """(code length per symbol, table log, [4 stream byte strings]) of a single-block level-1 frame, or None"""
    body = frame[4 + 1 + (4 if n >= 65536 + 256 else (2 if n >= 256 else 1)) + 3:]
    if (body[0] & 3) != 2:
        return None                                   # raw / rle literals
    sf = (body[0] >> 2) & 3
    if sf == 0:
        return None                                   # single stream: small sections, not the interesting case
    lh = 3 + (sf >= 2) + (sf == 3)
    v = int.from_bytes(body[:5], "little")
    comp = [(v >> 14) & 0x3FF, (v >> 14) & 0x3FF, (v >> 18) & 0x3FFF, (v >> 22) & 0x3FFFF][sf]
    lens = C.create_string_buffer(256)
    tl = orc.L.orc_dbg_huf_lengths(body[lh:lh + 260], 260, lens)
    hb = body[lh]
    hs = hb + 1 if hb < 128 else ((hb - 127 + 1) // 2) + 1
    js = lh + hs
    s = [int.from_bytes(body[js + 2 * i:js + 2 * i + 2], "little") for i in range(3)]
    s.append(comp - hs - 6 - sum(s))
    out, at = [], js + 6
    for ln in s:
        out.append(body[at:at + ln]); at += ln
    return list(lens.raw), tl, out


def decode_table(lens, tl):
    """nbBits for every tableLog-bit prefix (same cell order as HUF_readDTableX1: weight ascending, symbols ascending)"""
    nb = np.zeros(1 << tl, dtype=np.uint8)
    pos = 0
    for w in range(1, tl + 1):
        for s in range(256):
            if lens[s] and tl + 1 - lens[s] == w:
                nb[pos:pos + (1 << (w - 1))] = lens[s]; pos += 1 << (w - 1)
    assert pos == 1 << tl
    return nb


def study(stream, nb, tl, trials=200):
    bits = np.unpackbits(np.frombuffer(stream, dtype=np.uint8), bitorder="little")   # bit i of the stream (LSB-first bytes)
    top = len(bits) - 1
    while not bits[top]: top -= 1                     # the start mark
    # read position p counts bits below the mark still unread; a code is the tl bits just below the current position
    def peek(p):
        lo = p - tl
        if lo >= 0: chunk = bits[lo:p]
        else: chunk = np.concatenate([np.zeros(-lo, dtype=np.uint8), bits[0:p]])
        return int(chunk[::-1].dot(1 << np.arange(tl - 1, -1, -1))) if False else int("".join(map(str, chunk[::-1])), 2)
    true = set(); p = top
    while p > 0:
        true.add(p); p -= int(nb[peek(p)])
    res = []
    for _ in range(trials):
        q = int(rng.integers(tl, top)); start = q; syms = 0
        while q > 0 and q not in true:
            q -= int(nb[peek(q)]); syms += 1
        if q > 0: res.append((start - q, syms))
    return res, top


def main():
    rows = []
    for kind in sorted(a.corpus.KINDS):
        data = a.corpus.make(kind, per_kind * 131072, seed=3).tobytes()
        allb, alls = [], []
        for i in range(per_kind):
            part = data[i * 131072:(i + 1) * 131072]
            ls = literal_streams(orc.compress(ol.ZSTD, part), len(part))
            if ls is None: continue
            lens, tl, streams = ls
            nb = decode_table(lens, tl)
            for st in streams:
                if len(st) < 64: continue
                r, total_bits = study(st, nb, tl, trials=60)
                allb += [x[0] for x in r]; alls += [x[1] for x in r]
        if allb:
            b = np.array(allb); s = np.array(alls)
            rows.append((kind, len(b), float(np.median(b)), float(np.percentile(b, 90)), float(np.percentile(b, 99)), int(b.max()), float(np.median(s)), float(np.percentile(s, 99))))
    print("kind            trials  bits-to-sync median / p90 / p99 / max      symbols median / p99")
    for r in rows:
        print(f"{r[0]:14s} {r[1]:7d}  {r[2]:8.0f} {r[3]:6.0f} {r[4]:6.0f} {r[5]:6d}      {r[6]:6.0f} {r[7]:6.0f}")


if __name__ == "__main__":
    main()
