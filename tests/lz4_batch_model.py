"""CPU model of the LZ4 compressor's batches of 64 consecutive positions (csrc/lz4_kernels.hip lz4_compress_wave, fast path), lane
by lane in plain Python.  What is modelled is the DECISION logic of a batch: every lane reads the table as it was when the batch
began; a lane whose slot an earlier lane of the batch names (a duplicate) is decided when the walk arrives there - its candidate
is the position of the latest earlier lane on the slot whose table write has happened (as a probe, or as the ip-2 back-fill of
the match that covers it), else the table's entry; the walk goes from match to match; the commit writes every probed lane and
every back-fill, the latest writer of a slot owning it; what the next batch starts with (position, probes already made, whether
the ip-2 back-fill is still due).  Match lengths and catch-ups are exact here (the kernel's 28-byte windows and its
one-at-a-time path are not what this model is about), and whatever is not a fast-path batch (the chunk's last hundred bytes,
searches beyond 66 probes) goes through a plain sequential search.
TEST INFRASTRUCTURE: tests/test_lz4_batch_model.py compares the sequences with those of the oracle's compressed blocks."""

HASH_LOG = 12
PRIME5 = 889523592379
MFLIMIT, LASTLITERALS, MINMATCH, DIST_MAX = 12, 5, 4, 65535
M64 = (1 << 64) - 1


def hash5(src, p):
    v = int.from_bytes(src[p:p + 8].ljust(8, b"\0"), "little")
    return ((((v << 24) & M64) * PRIME5) & M64) >> (64 - HASH_LOG)


def hash4(src, p):
    """inputs below LZ4_64Klimit: a table of 16-bit entries, 13-bit hash of four bytes (lz4.c:759-768)"""
    v = int.from_bytes(src[p:p + 4].ljust(4, b"\0"), "little")
    return ((v * 2654435761) & 0xFFFFFFFF) >> (32 - (HASH_LOG + 1))


def count(src, a, b, limit):
    n = 0
    while a + n < limit and src[a + n] == src[b + n]:
        n += 1
    return n


def sched(i):
    """sum of the first i steps of the search schedule at acceleration 1 (lz4.c:1486-1510)"""
    if i <= 0: return 0
    n = 62 + i; q = n >> 6; r = n & 63
    return 1 + 32 * q * (q - 1) + q * (r + 1)


def parse_sequences(block: bytes):
    """(literal length, offset, match length) of every sequence of an LZ4 block; the last literals are left out"""
    out = []; i = 0; n = len(block)
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
        out.append((lit, off, ml + 4))
    return out


class Model:
    def __init__(self, src: bytes):
        self.src = src; self.n = len(src)
        self.u16 = self.n < 65547                                # LZ4_64Klimit (lz4.c:1384): no distance check either
        self.hash = hash4 if self.u16 else hash5
        self.T = [0] * (1 << (HASH_LOG + 1))
        self.ip = 1; self.anchor = 0; self.extra = False; self.i0 = 0
        self.seqs = []
        self.stats = dict(batches=0, dup_stops=0, dup_inbatch=0, dup_table=0, dup_nohit=0, sequential=0)

    # ---- one sequence the exact way (catch-up, extension, record) ----
    def take(self, m_pos, match):
        src = self.src
        while m_pos > self.anchor and match > 0 and src[m_pos - 1] == src[match - 1]:
            m_pos -= 1; match -= 1
        ml = MINMATCH + count(src, m_pos + MINMATCH, match + MINMATCH, self.n - LASTLITERALS)
        self.seqs.append((m_pos - self.anchor, m_pos - match, ml))
        return m_pos + ml                                       # where the parse goes on

    # ---- a search outside the fast path: the reference's loop from probe i0 of the search in progress ----
    def sequential_search(self):
        src, T, mfl1, hf = self.src, self.T, self.n - MFLIMIT + 1, self.hash
        self.stats["sequential"] += 1
        ex = 1 if self.extra else 0
        if self.extra and self.i0 == 0:
            T[hf(src, self.ip - 2)] = self.ip - 2              # :1750
        base = self.ip + 1 if self.extra else self.ip
        k = self.i0 - ex                                         # -1: the probe of ip itself behind a match (:1763-1800)
        while True:
            pos = base - 1 if k < 0 else base + sched(k)
            if base + sched(k + 1) > mfl1: return False          # forwardIp > mflimitPlusOne (:1513): last literals
            h = hf(src, pos); cand = T[h]; T[h] = pos
            if cand + DIST_MAX >= pos and src[cand:cand + 4] == src[pos:pos + 4]:
                self.ip = self.take(pos, cand); self.anchor = self.ip
                if self.ip >= mfl1: return False                 # :1747
                self.extra = True; self.i0 = 0
                return True
            k += 1

    # ---- a batch of 64 consecutive positions ----
    def batch(self):
        src, T, n, hf = self.src, self.T, self.n, self.hash
        mfl1 = n - MFLIMIT + 1
        self.stats["batches"] += 1
        ex = 1 if self.extra else 0
        s0 = self.ip + self.i0
        if self.extra and self.i0 == 0:
            T[hf(src, self.ip - 2)] = self.ip - 2              # the back-fill, before the batch reads the table
        own = min(64, 66 + ex - self.i0)
        h = [hf(src, s0 + L) for L in range(64)]
        ent = [T[h[L]] for L in range(64)]                       # the table as the batch found it
        my4 = [src[s0 + L:s0 + L + 4] for L in range(64)]
        tabok = [ent[L] + DIST_MAX >= s0 + L and src[ent[L]:ent[L] + 4] == my4[L] for L in range(64)]
        dup = [any(h[x] == h[L] for x in range(L)) for L in range(64)]
        marks = []                                               # (lane, end lane) of the batch's matches, in order
        written = []                                             # lanes whose table write happens, in time order

        def write_happened(x):
            for (m, e) in marks:
                if m < x < e:
                    return x == e - 2                            # inside a match: only the back-fill lane is written
            return True                                          # a probe (or a match's own lane)

        cur = 0                                                  # lane where the search in progress starts
        first_search = True
        done = False
        while True:
            hi = own if first_search else 64
            x = cur; hit_lane = None; cand_pos = 0
            while x < hi:
                if dup[x]:
                    self.stats["dup_stops"] += 1
                    il = None
                    for y in range(x - 1, -1, -1):
                        if h[y] == h[x] and write_happened(y): il = y; break
                    if il is not None:
                        if my4[il] == my4[x]: hit_lane, cand_pos = x, s0 + il; self.stats["dup_inbatch"] += 1
                        else: self.stats["dup_nohit"] += 1
                    elif tabok[x]: hit_lane, cand_pos = x, ent[x]; self.stats["dup_table"] += 1
                    else: self.stats["dup_nohit"] += 1
                elif tabok[x]:
                    hit_lane, cand_pos = x, ent[x]
                if hit_lane is not None: break
                x += 1
            if hit_lane is None:
                # lanes [cur, hi) were probed without a match
                written += list(range(cur, hi))
                if first_search and not marks: self.i0 += own    # the search goes on in the next batch
                else: self.i0 = 64 - cur
                break
            written += list(range(cur, hit_lane + 1))
            nip = self.take(s0 + hit_lane, cand_pos)
            self.anchor = nip
            e = nip - s0
            marks.append((hit_lane, e))
            first_search = False
            self.ip = nip; self.extra = True; self.i0 = 0
            if nip >= mfl1: done = True; break
            if e >= 64: break                                    # the next batch starts behind this match (and back-fills ip-2)
            written.append(e - 2)                                # the back-fill of this match belongs to this batch
            cur = e
        # the commit: every write in time order - the latest writer of a slot owns it
        for L in written:
            T[h[L]] = s0 + L
        return not done

    def run(self):
        n = self.n
        if n < 13: return self.seqs
        mfl1 = n - MFLIMIT + 1
        alive = True
        while alive:
            ex = 1 if self.extra else 0
            s0 = self.ip + self.i0
            if self.i0 <= 65 + ex and s0 + 92 <= mfl1:
                # (a batch that ended inside its 64 lanes after a match leaves i0 = 64 - cur probes of the next search made
                #  and the back-fill done: extra stays set, i0 != 0)
                alive = self.batch()
            else:
                alive = self.sequential_search()
        return self.seqs
