"""CPU model of the LZ4 decoder's group path (csrc/lz4_kernels.hip lz4_decompress_wave + csrc/lz_group.hip.h), lane by lane in
plain Python: the scan of 64 byte positions per round with its walk, the 4-byte records, the group execution through a 2 KB
block with 64 bytes of history (fit / carry-over / forced steps, far and in-block copies, block out) and the hand-over to the
one-sequence path.  TEST INFRASTRUCTURE: tests/test_lz4_group_model.py checks it against the oracle on the CPU, so the
bookkeeping of the kernel (what is buffered, where a group ends, where the token chain resumes) is validated without a GPU."""

K_RING, K_RECS, K_BLOCK, K_HIST = 2048, 64, 2048, 64


class Malformed(Exception):
    pass


def decode_one(src, ip, out, oend):
    """one sequence by the rules of lz4.c:3563-3790 (the kernel's lz4_decode_one); returns (ip, done)"""
    iend = len(src)
    token = src[ip]; ip += 1
    ln = token >> 4
    if ln == 15:
        if ip >= iend - 15: raise Malformed
        while True:
            s = src[ip]; ip += 1; ln += s
            if ip >= iend - 15: break
            if s != 255: break
    if len(out) + ln > oend - 12 or ip + ln > iend - 8:
        if ip + ln != iend or len(out) + ln > oend: raise Malformed
        out += src[ip:ip + ln]
        return ip + ln, True
    out += src[ip:ip + ln]; ip += ln
    off = src[ip] | (src[ip + 1] << 8); ip += 2
    ml = token & 15
    if ml == 15:
        while True:
            s = src[ip]; ip += 1; ml += s
            if ip >= iend - 5 + 1: raise Malformed
            if s != 255: break
    ml += 4
    if off > len(out): raise Malformed
    if len(out) + ml > oend - 5: raise Malformed
    if off == 0:
        out += bytes(ml)
    else:
        for _ in range(ml): out.append(out[-off])
    return ip, False


def decode(src: bytes, cap: int, trace=None):
    """returns the decoded bytes (raises Malformed as the kernel returns < 0)"""
    n = len(src)
    if cap == 0:
        if n == 1 and src[0] == 0: return b""
        raise Malformed
    if n <= 0: raise Malformed
    out = bytearray()
    oend = cap
    ip = 0; rend = 0                      # the ring is modelled as "bytes [.., rend) are staged"
    recs = []; gbase = 0
    stats = dict(rounds=0, groups=0, seq=0, forced=0, carried=0, steps=0)
    while True:
        # staging (the model keeps the kernel's arithmetic, not its timing)
        if ip > rend: rend = ip & ~15
        while rend - ip < 640 and rend + 1024 <= n: rend += 1024
        why = 0 if len(recs) > K_RECS - 22 else 3
        while len(recs) <= K_RECS - 22 and rend - ip >= 352:
            stats["rounds"] += 1
            nx = []; rec = []
            for lane in range(64):
                p = ip + lane
                t = src[p]; e1 = src[p + 1] if p + 1 < n else 0
                lit = t >> 4; x1 = lit == 15
                hard = x1 and e1 > 240
                if x1: lit += e1
                q = p + 1 + (1 if x1 else 0) + lit
                w2 = [src[q + j] if q + j < n else 0 for j in range(3)]
                off = w2[0] | (w2[1] << 8); e2 = w2[2]
                x2 = (t & 15) == 15
                mcode = (t & 15) + (e2 if x2 else 0)
                hard = hard or (x2 and e2 > 240)
                nx.append(-1 if hard else q + 2 + (1 if x2 else 0) - ip)
                rec.append((off, lit, mcode))
            cur = 0; stop = 0; took = []
            while cur < 64:
                v = nx[cur]
                if v < 0: stop = 2; break
                took.append(cur); cur = v
            why = stop
            if took:
                if not recs: gbase = ip
                for l in took: recs.append(rec[l])
                ip += cur
            assert len(recs) <= K_RECS, len(recs)
            if why != 0 or rend - ip < 640: break
        if why == 0 and len(recs) <= K_RECS - 22: continue
        forced = False
        while recs and not forced:
            stats["groups"] += 1
            nrec = len(recs)
            tok = []; lp = []; mst = []; oinc = []; c = gbase; o = 0
            for (off, lit, mcode) in recs:
                mlen = mcode + 4
                tok.append(c); c += 3 + lit + (1 if lit >= 15 else 0) + (1 if mlen >= 19 else 0)
                lp.append(o); mst.append(o + lit); o += lit + mlen; oinc.append(o)
            op = len(out)
            room = oend - 32 - op
            capo = 0 if room <= 0 else min(room, K_BLOCK)
            ng = sum(1 for k in range(nrec) if oinc[k] <= capo)
            fb = 64
            for k, (off, lit, mcode) in enumerate(recs):
                mlen = mcode + 4; srel = mst[k] - off
                if off == 0 or off > op + mst[k] or (srel + mlen > 0 and srel < -K_HIST): fb = k; break
            if fb < ng: ng = fb; forced = True
            elif ng == 0: forced = True
            if forced:
                stats["forced"] += 1
                ip = tok[ng]
                if rend - ip > K_RING: rend = ip & ~15
            if ng:
                total = oinc[ng - 1]
                hist = bytes(out[-K_HIST:]).rjust(K_HIST, b"\0")
                blk = bytearray(K_BLOCK)
                cb = bytearray(hist) + blk                     # cb[64 + r] = block-relative r
                for k in range(ng):
                    off, lit, mcode = recs[k]; mlen = mcode + 4
                    ls = tok[k] + 1 + (1 if lit >= 15 else 0)
                    cb[K_HIST + lp[k]:K_HIST + lp[k] + lit] = src[ls:ls + lit]
                    srel = mst[k] - off
                    if srel + mlen <= 0:                       # far: from the output in front of the block
                        s0 = op + srel
                        assert s0 >= 0
                        cb[K_HIST + mst[k]:K_HIST + mst[k] + mlen] = out[s0:s0 + mlen]
                for k in range(ng):                            # in-block copies, in order
                    off, lit, mcode = recs[k]; mlen = mcode + 4
                    srel = mst[k] - off
                    if srel + mlen > 0:
                        assert srel >= -K_HIST
                        for i in range(mlen): cb[K_HIST + mst[k] + i] = cb[K_HIST + srel + (i if off >= mlen else i % off)] if off < mlen else cb[K_HIST + srel + i]
                out += cb[K_HIST:K_HIST + total]
                stats["seq"] += ng
            if forced: recs = []
            else:
                left = nrec - ng
                if left: stats["carried"] += 1
                gbase = tok[ng] if ng < nrec else tok[nrec - 1] + (3 + recs[-1][1] + (1 if recs[-1][1] >= 15 else 0) + (1 if recs[-1][2] + 4 >= 19 else 0))
                recs = recs[ng:]
                if why == 0: break
        if why == 0 and not forced: continue
        stats["steps"] += 1
        ip, done = decode_one(src, ip, out, oend)
        if done: break
    if trace is not None: trace.update(stats)
    return bytes(out)
