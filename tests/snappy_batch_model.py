"""CPU model of the Snappy compressor's batches of 64 consecutive positions (csrc/snappy_kernels.hip sn_compress_fragment, the
batch that follows a copy), lane by lane in plain Python.  As tests/lz4_batch_model.py does for LZ4, it states the DECISION rule:
every lane reads the table as it was when the batch began; a lane whose slot an earlier lane of the batch names (a duplicate) is
decided when the walk arrives there - its candidate is the position of the latest earlier lane on the slot whose table write
has happened (a probe, or the ip-1 back-fill of the copy over it, snappy.cc:1016-1017), else the table's entry; a search owns 34
lanes (the re-probe of ip and the 33 one-byte steps of the schedule, snappy.cc:935-974); the commit writes every probed lane and
every back-fill, the latest writer of a slot owning it.  Whatever is not such a batch (the fragment's start, searches beyond 34
probes, the last hundred bytes) goes through a plain sequential search.
TEST INFRASTRUCTURE: tests/test_snappy_batch_model.py compares the copies with those of the oracle's compressed streams."""

MARGIN, BLOCK, MIN_TABLE, MAX_TABLE, OWN = 15, 65536, 256, 1 << 14, 34


def rd32(b, p):
    return int.from_bytes(b[p:p + 4], "little")


def parse_elements(stream: bytes):
    """(uncompressed length, [(literal length, offset, copy length)]) of a Snappy stream; elements of one copy (EmitCopy cuts a
    long one into pieces, snappy.cc:507-568) are joined again: consecutive copy elements with one offset and no literal between;
    literals that no copy follows (the end of a fragment) are dropped"""
    i = 0; n = 0; sh = 0
    while True:
        b = stream[i]; i += 1; n |= (b & 127) << sh; sh += 7
        if b < 128: break
    out = []; lit = 0; last_was_copy = False
    while i < len(stream):
        t = stream[i]; kind = t & 3
        if kind == 0:
            ln = (t >> 2) + 1; i += 1
            if ln > 60:
                nb = ln - 60; ln = int.from_bytes(stream[i:i + nb], "little") + 1; i += nb
            i += ln; lit += ln; last_was_copy = False
            continue
        if kind == 1: ln = 4 + ((t >> 2) & 7); off = ((t >> 5) << 8) | stream[i + 1]; i += 2
        elif kind == 2: ln = (t >> 2) + 1; off = stream[i + 1] | (stream[i + 2] << 8); i += 3
        else: ln = (t >> 2) + 1; off = int.from_bytes(stream[i + 1:i + 5], "little"); i += 5
        if last_was_copy and out[-1][1] == off: out[-1] = (out[-1][0], off, out[-1][2] + ln)
        else: out.append((lit, off, ln))
        lit = 0; last_was_copy = True
    return n, out


class Fragment:
    def __init__(self, src: bytes):
        self.src = src; n = self.n = len(src)
        self.tsize = MAX_TABLE if n > MAX_TABLE else (MIN_TABLE if n < MIN_TABLE else 2 << ((n - 1).bit_length() - 1))
        self.shift = 32 - (self.tsize.bit_length() - 1)
        self.T = [0] * self.tsize
        self.ip = 0; self.anchor = 0; self.extra = False; self.i0 = 0
        self.copies = []
        self.stats = dict(batches=0, dup_stops=0, dup_inbatch=0, dup_table=0, dup_nohit=0, sequential=0)

    def hash(self, p):
        return ((rd32(self.src, p) * 0x1e35a7bd) & 0xFFFFFFFF) >> self.shift

    def take(self, pos, cand):
        src, n = self.src, self.n
        m = 4
        while pos + m < n and src[cand + m] == src[pos + m]: m += 1
        self.copies.append((pos - self.anchor, pos - cand, m))
        self.anchor = pos + m
        return pos + m

    # ---- the reference's loops from probe i0 of the search in progress (snappy.cc:903-1032) ----
    def sequential_search(self):
        src, T, ip_limit = self.src, self.T, self.n - MARGIN
        self.stats["sequential"] += 1
        ex = 1 if self.extra else 0
        if self.extra and self.i0 == 0: T[self.hash(self.ip - 1)] = self.ip - 1          # :1016
        pos = self.ip + 1; skip = 32                                                       # next_emit = ip++
        k = -ex
        while True:
            if k < 0: p = self.ip; nxt = pos                                               # the re-probe of ip in the emit_match loop
            else:
                step = skip >> 5; skip += step
                p = pos; nxt = pos + step
                if nxt > ip_limit: return False                                            # emit_remainder
            if k + ex >= self.i0:                                                          # (earlier probes were made by a batch)
                h = self.hash(p); cand = T[h]; T[h] = p
                if rd32(src, cand) == rd32(src, p):
                    self.ip = self.take(p, cand)
                    if self.ip >= ip_limit: return False
                    self.extra = True; self.i0 = 0
                    return True
            if k >= 0: pos = nxt
            k += 1

    # ---- a batch of 64 consecutive positions behind a copy ----
    def batch(self):
        src, T, ip_limit = self.src, self.T, self.n - MARGIN
        self.stats["batches"] += 1
        s0 = self.ip
        T[self.hash(s0 - 1)] = s0 - 1                                                      # the back-fill, before the batch reads the table
        h = [self.hash(s0 + L) for L in range(64)]
        ent = [T[h[L]] for L in range(64)]                                                 # the table as the batch found it
        my4 = [rd32(src, s0 + L) for L in range(64)]
        tabok = [rd32(src, ent[L]) == my4[L] for L in range(64)]
        dup = [any(h[x] == h[L] for x in range(L)) for L in range(64)]
        marks = []; written = []

        def write_happened(x):
            for (m, e) in marks:
                if m < x < e: return x == e - 1                                            # inside a copy: only the back-fill lane
            return True

        cur = 0
        while True:
            hi = min(64, cur + OWN)
            x = cur; hit = None; cand = 0
            while x < hi:
                if dup[x]:
                    self.stats["dup_stops"] += 1
                    il = next((y for y in range(x - 1, -1, -1) if h[y] == h[x] and write_happened(y)), None)
                    if il is not None:
                        if my4[il] == my4[x]: hit, cand = x, s0 + il; self.stats["dup_inbatch"] += 1
                        else: self.stats["dup_nohit"] += 1
                    elif tabok[x]: hit, cand = x, ent[x]; self.stats["dup_table"] += 1
                    else: self.stats["dup_nohit"] += 1
                elif tabok[x]: hit, cand = x, ent[x]
                if hit is not None: break
                x += 1
            if hit is None:
                written += list(range(cur, hi))
                self.ip = s0 + cur; self.extra = True; self.i0 = hi - cur
                alive = True
                break
            written += list(range(cur, hit + 1))
            nip = self.take(s0 + hit, cand)
            e = nip - s0
            marks.append((hit, e))
            self.ip = nip; self.extra = True; self.i0 = 0
            if nip >= ip_limit: alive = False; break
            if e >= 64: alive = True; break                                                # the next batch back-fills ip-1
            written.append(e - 1)
            cur = e
        for L in written: T[h[L]] = s0 + L                                                 # in time order: the latest writer owns a slot
        return alive

    def run(self):
        if self.n < MARGIN: return self.copies
        ip_limit = self.n - MARGIN
        alive = True
        while alive:
            if self.extra and self.i0 == 0 and self.ip + 64 + 40 <= ip_limit: alive = self.batch()
            else: alive = self.sequential_search()
        return self.copies

