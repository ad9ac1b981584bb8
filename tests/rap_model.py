"""TEST INFRASTRUCTURE: a pure-Python restatement of the reference's RAP join, used to check the GPU pack
(csrc/rap.hip) on partition geometries the threaded reference build cannot be steered into.

  frame   threads/threads.h:20-112, threads/threads.c:104-111 : 'AOCL_LLC' | u32 length | u32 partitions | entries
  LZ4     lz4.c:2736-2906 : partition k loses its final literal run, which is merged into the first token of
          partition k+1; all-literal partitions (:2808-2823) are carried on
  Snappy  snappy.cc:2567-2654 : one varint for the total, partitions without theirs
  Zstd    zstd_compress.c:5343-5385,5499-5517 : frame inside a skippable frame, partitions are whole frames
The per-partition payloads come from the oracle's one-shot compressors (the MT compressors make the same parse,
lz4.c:1853-2351 vs :1371-1845).  It is pinned against the real threaded build in tests/test_rap_model.py."""
import struct

import oracle_lib as ol

MAGIC = 0x434C4C5F4C434F41


def _lz4_last_run(c: bytes):
    """(length of the final literal run, offset of its token) of one LZ4 block."""
    ip = 0
    while True:
        t = c[ip]; start = ip; ip += 1
        lit = t >> 4
        if lit == 15:
            while True:
                b = c[ip]; ip += 1; lit += b
                if b != 255:
                    break
        ip += lit
        if ip >= len(c):
            return lit, start
        ip += 2
        if (t & 15) == 15:
            while True:
                b = c[ip]; ip += 1
                if b != 255:
                    break


def _varint(v: int) -> bytes:
    out = bytearray()
    while v >= 128:
        out.append((v & 0x7F) | 0x80); v >>= 7
    out.append(v)
    return bytes(out)


def pack(oracle, codec: int, data: bytes, chunk: int) -> bytes:
    parts = [data[i:i + chunk] for i in range(0, len(data), chunk)] or [b""]
    K = len(parts)
    comp = [oracle.compress(codec, p) for p in parts]
    if K == 1:
        return comp[0]                                             # one partition: the plain stream
    base = 8 if codec == ol.ZSTD else 0
    frame = 16 + 12 * K
    ents, body = [], bytearray()
    off = frame                                                     # offsets count from the magic word
    if codec == ol.ZSTD:
        for p, c in zip(parts, comp):
            ents.append((off, len(c), len(p))); off += len(c); body += c
    elif codec == ol.SNAPPY:
        v = _varint(len(data)); body += v; off += len(v)
        for p, c in zip(parts, comp):
            w = len(_varint(len(p)))
            ents.append((off, len(c) - w, len(p))); off += len(c) - w; body += c[w:]
    else:
        pend, pos = 0, 0
        for k, (p, c) in enumerate(zip(parts, comp)):
            last = k == K - 1
            run, start = _lz4_last_run(c)
            bend = len(c) if last else start
            L = 0 if last else run
            if k == 0:
                piece, dl = c[:bend], len(p) - L
            elif not last and bend == 0:                           # all literals: they travel on
                ents.append((off, 0, 0)); pend += len(p); pos += len(p)
                continue
            else:
                t = c[0]; cur = t >> 4; hi = 1
                if cur == 15:
                    while True:
                        e = c[hi]; hi += 1; cur += e
                        if e != 255:
                            break
                nl = cur + pend
                h = bytearray([(min(nl, 15) << 4) | (t & 15)])
                if nl >= 15:
                    x = nl - 15
                    while x >= 255:
                        h.append(255); x -= 255
                    h.append(x)
                piece = bytes(h) + data[pos - pend:pos] + c[hi:bend]
                dl = len(p) - L + pend
            ents.append((off, len(piece), dl)); off += len(piece); body += piece
            pend = L; pos += len(p)
    hdr = struct.pack("<qII", MAGIC, frame, K) + b"".join(struct.pack("<III", *e) for e in ents)
    if codec == ol.ZSTD:
        hdr = struct.pack("<II", 0x184D2A50, frame) + hdr
    return hdr + bytes(body)
