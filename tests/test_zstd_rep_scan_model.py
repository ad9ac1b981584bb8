"""The Zstd decoder resolves the offset history of a tile of 64 sequences by a prefix scan of "history maps" (csrc/zstd_kernels.hip:
RepMap, rep_compose, rep_entry, rep_resolve) instead of the reference's serial rule (ZSTD_decodeSequence, lib/decompress/
zstd_decompress_block.c:1370-1400: offset values 1..3 are repeat codes, shifted by one when the literal length is 0; value 3 with
literal length 0 is "first entry minus one", forced to 1 when that is 0).  This is the CPU model of that algebra: the same packed
representation (three entries of {kind, k, constant}, nine bits of meta per entry, identity = 0), composed by a Hillis-Steele scan,
against the serial rule - on random tiles that are rich in repeat codes, zero literal lengths and tiny history entries (the clamp)."""
import random


def serial(vals, lls, R):
    r0, r1, r2 = R
    outs = []
    for val, ll in zip(vals, lls):
        ll0 = 1 if ll == 0 else 0
        is_rep = val <= 3
        idx = val - 1 + ll0
        tt = r1 if idx == 1 else (r2 if idx == 2 else (r0 - 1) & 0xFFFFFFFF)
        if tt == 0:
            tt = 1
        off = (r0 if idx == 0 else tt) if is_rep else val - 3
        if (not is_rep) or idx >= 2:
            r2 = r1
        if not (is_rep and idx == 0):
            r1 = r0
            r0 = off
        outs.append(off)
    return outs, (r0, r1, r2)


def make_map(val, ll):
    is_rep = val <= 3
    idx = val - 1 + (1 if ll == 0 else 0)
    c = [(val - 3) & 0xFFFFFFFF, 0, 0]
    if not is_rep:
        meta = 3 | 3 << 9 | 3 << 18
    elif idx == 0:
        meta = 0
    elif idx == 1:
        meta = 1 | 3 << 9
    elif idx == 2:
        meta = 2 | 3 << 9 | 3 << 18
    else:
        meta = (1 << 2) | 3 << 9 | 3 << 18
    return (meta, c)


def compose(E, M):                      # first E, then M
    meta = 0
    c = [0, 0, 0]
    for j in range(3):
        fm = (M[0] >> (9 * j)) & 0x1FF
        km = ((fm & 3) + j) & 3
        fe = (E[0] >> (9 * (km & 3))) & 0x1FF
        ke = ((fe & 3) + km) & 3
        kk = (fe >> 2) + (fm >> 2)
        own = km == 3
        f = fm if own else (((ke - j) & 3) | (kk << 2))
        c[j] = M[1][j] if own else E[1][km if km < 3 else 0]
        meta |= f << (9 * j)
    return (meta, c)


def entry(A, j, R):
    f = (A[0] >> (9 * j)) & 0x1FF
    kind = ((f & 3) + j) & 3
    k = f >> 2
    base = A[1][j] if kind == 3 else R[kind]
    return (base - k if base > k else 1) if k else base


def scan(vals, lls, R):
    n = len(vals)
    agg = [make_map(v, l) for v, l in zip(vals, lls)] + [(0, [0, 0, 0])] * (64 - n)
    d = 1
    while d < 64:
        agg = [agg[i] if i < d else compose(agg[i - d], agg[i]) for i in range(64)]
        d *= 2
    return [entry(agg[i], 0, R) for i in range(n)], tuple(entry(agg[63], j, R) for j in range(3))


def test_identity_is_zero_and_neutral():
    ident = (0, [0, 0, 0])
    for val, ll in ((1, 5), (1, 0), (2, 0), (3, 0), (3, 7), (900, 0)):
        m = make_map(val, ll)
        for R in ((1, 4, 8), (7, 7, 7), (1, 1, 1)):
            assert [entry(compose(ident, m), j, R) for j in range(3)] == [entry(m, j, R) for j in range(3)]
            assert [entry(compose(m, ident), j, R) for j in range(3)] == [entry(m, j, R) for j in range(3)]


def test_scan_equals_the_serial_rule():
    rng = random.Random(20260105)
    for it in range(6000):
        n = rng.randint(1, 64)
        rich = rng.random()
        vals, lls = [], []
        for _ in range(n):
            if rng.random() < (0.85 if rich < 0.5 else 0.3):
                vals.append(rng.randint(1, 3))
            else:
                vals.append(rng.choice([4, 5, 6, 7, 100, 70000, 2 ** 31 + 5]))
            lls.append(0 if rng.random() < 0.45 else rng.randint(1, 9))
        R = (rng.choice([1, 2, 3, 4, 8, 1000]), rng.choice([1, 2, 4, 77]), rng.choice([1, 8, 3, 9999]))
        assert scan(vals, lls, R) == serial(vals, lls, R), (it, vals, lls, R)


def test_chains_of_minus_one_hit_the_clamp():
    # 64 x "offset value 3 after a zero literal length": first entry minus one every time, forced to 1 from the 8th on
    vals, lls, R = [3] * 64, [0] * 64, (8, 4, 1)
    got, end = scan(vals, lls, R)
    assert (got, end) == serial(vals, lls, R)
    assert got[:7] == [7, 6, 5, 4, 3, 2, 1] and set(got[7:]) == {1}
