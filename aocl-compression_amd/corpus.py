"""Deterministic synthetic corpora ("Silesia-shaped") for parity tests and bench.py.

Silesia itself is not available (no network), so BASELINE.json's inputs are synthesised from a
seeded numpy PCG64 stream.  The mix follows the public description of the Silesia members
(SURVEY.md section 8d): natural-language text / markup, source-like and log/JSON-like records,
executable-like binary, database-like fixed records, 16-bit medical-image-like samples and a
small incompressible tail.  Everything is rendered through one vectorised "token stream ->
bytes" gather, so 200 MB takes a few seconds on one core.

Nothing here is on the product path: it only feeds inputs to the codecs.
"""
from __future__ import annotations

import numpy as np

__all__ = ["make", "silesia_mix", "KINDS", "MIX_SHARES"]

_LETTERS = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
_LETTER_P = np.array([12.7, 9.1, 8.2, 7.5, 7.0, 6.7, 6.3, 6.1, 6.0, 4.3, 4.0, 2.8, 2.8, 2.4, 2.4, 2.2,
                      2.0, 2.0, 1.9, 1.5, 1.0, 0.8, 0.15, 0.15, 0.1, 0.07])
_LETTER_P = _LETTER_P / _LETTER_P.sum()


class _Vocab:
    """A list of byte strings stored as one flat array + offsets (for the vectorised renderer)."""

    def __init__(self, items):
        self.lens = np.array([len(x) for x in items], dtype=np.int64)
        self.off = np.zeros(len(items) + 1, dtype=np.int64)
        np.cumsum(self.lens, out=self.off[1:])
        self.data = np.frombuffer(b"".join(items), dtype=np.uint8)

    def __len__(self):
        return len(self.lens)


def _render(v: _Vocab, ids: np.ndarray, block: int = 1 << 21) -> np.ndarray:
    """Concatenate v[ids[0]] + v[ids[1]] + ... without a Python loop over tokens."""
    outs = []
    for s in range(0, len(ids), block):
        t = ids[s:s + block]
        lens = v.lens[t]
        total = int(lens.sum())
        if total == 0:
            continue
        ends = np.cumsum(lens)
        delta = np.repeat(v.off[t] - (ends - lens), lens)
        delta += np.arange(total, dtype=np.int64)
        outs.append(v.data[delta])
    return np.concatenate(outs) if outs else np.zeros(0, np.uint8)


def _zipf_ids(rng, n, size, s=1.05):
    """Zipf(s) over ranks 0..n-1 by inverse CDF (stable across numpy versions)."""
    w = 1.0 / np.power(np.arange(1, n + 1, dtype=np.float64), s)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    return np.searchsorted(cdf, rng.random(size), side="left").astype(np.int64)


def _words(rng, n, lo=2, hi=12):
    lens = rng.integers(lo, hi + 1, size=n)
    # frequent words tend to be short: sort lengths ascending over rank with heavy jitter
    lens = np.sort(lens)[np.argsort(np.argsort(np.log1p(np.arange(n)) + rng.normal(0, 2.5, n)))]
    letters = _LETTERS[np.searchsorted(np.cumsum(_LETTER_P), rng.random(int(lens.sum())))
                       .clip(0, len(_LETTERS) - 1)]
    out, p = [], 0
    for ln in lens.tolist():
        out.append(letters[p:p + ln].tobytes())
        p += ln
    return out


def _text(rng, nbytes):
    """enwik-style text: Zipf words, reused phrases, wiki-ish markup tokens."""
    nv = 50000
    words = _words(rng, nv)
    markup = [b"[[", b"]]", b"''", b"'''", b"==", b"{{", b"}}", b"|", b"<ref>", b"</ref>", b"&quot;", b"\n\n",
              b"\n* ", b"\n# ", b"<br />", b"&amp;", b"http://www.", b".org/", b".com/", b"[[Category:",
              b"[[Image:", b"|thumb|", b"px|", b"<!--", b"-->", b"&lt;", b"&gt;", b"</text>\n", b"<page>\n",
              b"    <title>", b"</title>\n", b"    <id>", b"</id>\n", b"      <timestamp>2006-0", b"Z</timestamp>\n"]
    seps = [b" ", b" ", b" ", b" ", b", ", b". ", b" ", b" ", b"; ", b" ", b".\n", b" ("]
    closers = [b") "]
    digits = [str(i).encode() for i in range(0, 2100)]
    items = words + markup + seps + closers + digits
    v = _Vocab(items)
    o_mark, o_sep = nv, nv + len(markup)
    o_dig = o_sep + len(seps) + len(closers)
    # phrase table: each phrase = a short sequence of (word, sep) pairs
    nph = 40000
    ph_len = rng.integers(2, 10, size=nph)
    ph_off = np.zeros(nph + 1, np.int64)
    np.cumsum(ph_len * 2, out=ph_off[1:])
    ph_tok = np.empty(int(ph_off[-1]), np.int64)
    ph_tok[0::2] = _zipf_ids(rng, nv, len(ph_tok) // 2, 1.02)
    ph_tok[1::2] = o_sep + rng.integers(0, len(seps), len(ph_tok) // 2)
    pv = _Vocab([b"x"])  # reuse renderer machinery on token ids: render phrases to token streams
    pv.lens, pv.off, pv.data = ph_len * 2, ph_off, ph_tok
    est_units = int(nbytes / 14) + 1024
    kind = rng.random(est_units)
    unit_ids = []
    # 58% phrases, 33% fresh words, 6% markup, 3% numbers
    n_ph = int((kind < 0.58).sum())
    phr = _zipf_ids(rng, nph, n_ph, 1.12)
    # assemble the per-unit token streams
    toks_ph = _render_tokens(pv, phr)
    n_w = int(((kind >= 0.58) & (kind < 0.91)).sum())
    toks_w = np.empty(n_w * 2, np.int64)
    toks_w[0::2] = _zipf_ids(rng, nv, n_w, 1.0)
    toks_w[1::2] = o_sep + rng.integers(0, len(seps), n_w)
    n_m = int(((kind >= 0.91) & (kind < 0.97)).sum())
    toks_m = o_mark + _zipf_ids(rng, len(markup), n_m, 0.8)
    n_d = est_units - n_ph - n_w - n_m
    toks_d = np.empty(n_d * 2, np.int64)
    toks_d[0::2] = o_dig + _zipf_ids(rng, len(digits), n_d, 0.7)
    toks_d[1::2] = o_sep
    # interleave the four streams deterministically by shuffling segment order
    segs = _interleave(rng, [(toks_ph, ph_len[phr] * 2), (toks_w, np.full(n_w, 2)), (toks_m, np.full(n_m, 1)),
                             (toks_d, np.full(n_d, 2))])
    del unit_ids
    out = _render(v, segs)
    return _fit(out, nbytes, rng)


def _render_tokens(pv, ids):
    lens = pv.lens[ids]
    total = int(lens.sum())
    ends = np.cumsum(lens)
    idx = np.repeat(pv.off[ids] - (ends - lens), lens) + np.arange(total, dtype=np.int64)
    return pv.data[idx]


def _interleave(rng, streams):
    """Merge several (tokens, unit_lengths) streams into one, unit by unit, in a random order."""
    unit_src = np.concatenate([np.full(len(ul), i, np.int64) for i, (_, ul) in enumerate(streams)])
    perm = rng.permutation(len(unit_src))
    unit_src = unit_src[perm]
    all_tok = np.concatenate([t for t, _ in streams])
    base, starts, lens = 0, [], []
    for t, ul in streams:
        e = np.cumsum(ul)
        starts.append(base + e - ul)
        lens.append(ul)
        base += len(t)
    starts = np.concatenate(starts)[perm]
    lens = np.concatenate(lens)[perm]
    total = int(lens.sum())
    ends = np.cumsum(lens)
    idx = np.repeat(starts - (ends - lens), lens) + np.arange(total, dtype=np.int64)
    return all_tok[idx]


def _fit(arr, nbytes, rng):
    if len(arr) >= nbytes:
        return np.ascontiguousarray(arr[:nbytes])
    reps = -(-nbytes // max(1, len(arr)))
    return np.ascontiguousarray(np.tile(arr, reps)[:nbytes])


def _jsonlog(rng, nbytes):
    """JSON / log records: templated lines with Zipf keys, timestamps, counters, hex ids."""
    hex2 = [b"%02x" % i for i in range(256)]
    d2 = [b"%02d" % i for i in range(100)]
    d3 = [b"%03d" % i for i in range(1000)]
    levels = [b"INFO", b"INFO", b"INFO", b"DEBUG", b"WARN", b"ERROR", b"TRACE", b"INFO"]
    svcs = [b"auth", b"gateway", b"billing", b"search", b"ingest", b"scheduler", b"storage", b"frontend",
            b"metrics", b"notifier", b"cache", b"db-proxy"]
    msgs = [b"request completed", b"connection accepted", b"cache miss", b"cache hit", b"retrying upstream",
            b"slow query detected", b"user session refreshed", b"payload validated", b"rate limit exceeded",
            b"heartbeat ok", b"flushing write buffer", b"compaction finished", b"lease renewed",
            b"token expired", b"permission denied for resource", b"queue depth above threshold"]
    paths = [b"/api/v1/users", b"/api/v1/orders", b"/api/v2/search", b"/healthz", b"/api/v1/items/",
             b"/static/app.js", b"/api/v1/login", b"/api/v1/cart", b"/metrics", b"/api/v2/recommend"]
    frag = [b'{"ts":"2024-', b"-", b"T", b":", b":", b".", b'Z","level":"', b'","svc":"', b"-", b'","msg":"',
            b'","path":"', b'","status":', b',"bytes":', b',"dur_ms":', b',"trace":"', b'","host":"node-',
            b'.dc', b'.example.net"}\n', b'","user":"u', b",\"tags\":[\"", b"\",\"", b"\"]"]
    status = [b"200", b"200", b"200", b"200", b"201", b"204", b"301", b"304", b"400", b"401", b"404", b"500", b"503"]
    groups = [hex2, d2, d3, levels, svcs, msgs, paths, frag, status]
    base, items = [], []
    for g in groups:
        base.append(len(items))
        items += g
    HEX, D2, D3, LV, SV, MS, PA, FR, ST = base
    v = _Vocab(items)
    n = int(nbytes / 215) + 64
    z = lambda k, s=1.0: _zipf_ids(rng, k, n, s)
    t = np.cumsum(rng.integers(0, 40, n))  # ms clock
    sec = (t // 1000)
    cols = [
        np.full(n, FR + 0), D2 + 1 + (sec // 2592000) % 12, np.full(n, FR + 1), D2 + 1 + (sec // 86400) % 28,
        np.full(n, FR + 2), D2 + (sec // 3600) % 24, np.full(n, FR + 3), D2 + (sec // 60) % 60, np.full(n, FR + 4),
        D2 + sec % 60, np.full(n, FR + 5), D3 + t % 1000, np.full(n, FR + 6), LV + z(len(levels), 0.3),
        np.full(n, FR + 7), SV + z(len(svcs)), np.full(n, FR + 8), D2 + rng.integers(0, 16, n),
        np.full(n, FR + 9), MS + z(len(msgs)), np.full(n, FR + 10), PA + z(len(paths)),
        np.full(n, FR + 18), D3 + z(1000, 0.9), D2 + rng.integers(0, 100, n),
        np.full(n, FR + 11), ST + z(len(status), 0.2), np.full(n, FR + 12), D3 + rng.integers(1, 1000, n),
        D2 + rng.integers(0, 100, n), np.full(n, FR + 13), D3 + z(1000, 1.2), np.full(n, FR + 14),
    ]
    cols += [HEX + rng.integers(0, 256, n) for _ in range(16)]
    cols += [np.full(n, FR + 15), D2 + z(64, 0.5), np.full(n, FR + 16), D2 + 1 + z(4, 0.5), np.full(n, FR + 17)]
    ids = np.stack(cols, axis=1).reshape(-1)
    return _fit(_render(v, ids), nbytes, rng)


def _binary(rng, nbytes):
    """Executable-like: Zipf 'instructions' with short immediates, tables of near pointers, zero padding."""
    nop = 600
    ops = []
    for i in range(nop):
        ln = int(rng.integers(1, 8))
        ops.append(bytes(rng.integers(0, 256, ln, dtype=np.uint8)))
    imm = [bytes([i]) for i in range(256)]
    pad = [b"\x00" * 16, b"\xcc" * 8, b"\x90" * 4, b"\x00\x00\x00\x00"]
    hi = [bytes([a, 0x40, 0x00]) for a in range(0, 256, 4)] + [b"\xff\xff\xff", b"\x00\x00\x00"]
    items = ops + imm + pad + hi
    v = _Vocab(items)
    o_imm, o_pad, o_hi = nop, nop + 256, nop + 256 + len(pad)
    n = int(nbytes / 3.4) + 64
    kind = rng.random(n)
    ids = _zipf_ids(rng, nop, n, 1.1)
    # runs of code re-used (function prologues etc.): copy earlier windows forward
    m_imm = (kind > 0.62) & (kind <= 0.80)
    ids[m_imm] = o_imm + (_zipf_ids(rng, 256, int(m_imm.sum()), 1.3) * 8) % 256
    m_hi = (kind > 0.80) & (kind <= 0.90)
    ids[m_hi] = o_hi + _zipf_ids(rng, len(hi), int(m_hi.sum()), 1.0)
    m_pad = kind > 0.985
    ids[m_pad] = o_pad + rng.integers(0, len(pad), int(m_pad.sum()))
    # replicate idioms: overwrite random windows with copies of earlier windows
    nrep = n // 14
    src = rng.integers(0, max(1, n - 64), nrep)
    dst = rng.integers(0, max(1, n - 64), nrep)
    ln = rng.integers(3, 40, nrep)
    for s, d, l in zip(src.tolist(), dst.tolist(), ln.tolist()):
        ids[d:d + l] = ids[s:s + l]
    return _fit(_render(v, ids), nbytes, rng)


def _database(rng, nbytes):
    """Fixed-width catalogue records (osdb / nci-like): many repeated fields, sorted keys."""
    rec = 64
    n = nbytes // rec + 1
    out = np.full((n, rec), 0x20, np.uint8)
    key = np.cumsum(rng.integers(1, 4, n)).astype(np.int64)
    dig = np.array([ord(c) for c in "0123456789"], np.uint8)
    for c in range(10):
        out[:, 9 - c] = dig[(key // 10 ** c) % 10]
    cats = _Vocab([b"ALPHA   ", b"BETA    ", b"GAMMA   ", b"DELTA   ", b"EPSILON ", b"ZETA    ", b"ETA     ", b"THETA   "])
    cid = _zipf_ids(rng, 8, n, 1.0)
    out[:, 11:19] = cats.data.reshape(8, 8)[cid]
    val = rng.integers(0, 100000, n)
    val[rng.random(n) < 0.6] = 0
    for c in range(6):
        out[:, 26 - c] = dig[(val // 10 ** c) % 10]
    flags = np.frombuffer(b"YN", np.uint8)[(rng.random((n, 6)) < 0.15).astype(np.int64)]
    out[:, 28:34] = flags
    names = _Vocab(_words(rng, 512, 4, 10))
    nid = _zipf_ids(rng, 512, n, 0.9)
    for j in range(10):
        ch = np.where(j < names.lens[nid], names.data[np.minimum(names.off[nid] + j, len(names.data) - 1)], 0x20)
        out[:, 36 + j] = ch
    out[:, 47:63] = np.frombuffer(b"0000-00-00 00:00", np.uint8)
    day = (key // 50) % 28 + 1
    out[:, 55] = dig[day // 10]
    out[:, 56] = dig[day % 10]
    out[:, 63] = 0x0A
    return _fit(out.reshape(-1), nbytes, rng)


def _image16(rng, nbytes):
    """12-bit samples in 16-bit little-endian words: smooth field + sensor noise (x-ray / MR-like)."""
    n = nbytes // 2 + 1
    w = 2048
    rows = -(-n // w)
    x = np.linspace(0, 9.0, w)
    y = np.linspace(0, 7.0, rows)[:, None]
    field = 1800 + 900 * np.sin(x)[None, :] * np.cos(y) + 500 * np.sin(0.37 * x[None, :] * (1 + 0.1 * y))
    noise = rng.normal(0, 5.5, (rows, w))
    img = np.clip(field + noise, 0, 4095).astype(np.uint16)
    img[:, : w // 10] = 0  # collimated border
    return _fit(img.reshape(-1).view(np.uint8), nbytes, rng)


def _random(rng, nbytes):
    return rng.integers(0, 256, nbytes, dtype=np.uint8)


KINDS = {"text": _text, "jsonlog": _jsonlog, "binary": _binary, "database": _database, "image16": _image16,
         "random": _random}

# approximate shares of the Silesia members by type
MIX_SHARES = (("text", 0.43), ("jsonlog", 0.12), ("binary", 0.26), ("database", 0.08), ("image16", 0.09),
              ("random", 0.02))


def make(kind: str, nbytes: int, seed: int = 0xA0C1) -> np.ndarray:
    """One corpus of `nbytes` bytes of the given kind (uint8 array)."""
    if nbytes <= 0:
        return np.zeros(0, np.uint8)
    rng = np.random.Generator(np.random.PCG64(seed))
    out = KINDS[kind](rng, int(nbytes))
    assert out.dtype == np.uint8 and len(out) == nbytes
    return out


def silesia_mix(nbytes: int, seed: int = 0xA0C1, segment: int = 4 << 20) -> np.ndarray:
    """Silesia-shaped mix: segments of `segment` bytes whose kinds follow MIX_SHARES."""
    if nbytes <= 0:
        return np.zeros(0, np.uint8)
    nseg = -(-nbytes // segment)
    rng = np.random.Generator(np.random.PCG64(seed ^ 0x51E51A))
    # deterministic kind order with the right proportions
    quota = [(k, s * nseg) for k, s in MIX_SHARES]
    kinds = []
    acc = {k: 0.0 for k, _ in MIX_SHARES}
    for i in range(nseg):
        k = max(quota, key=lambda q: q[1] * (i + 1) / nseg - acc[q[0]])[0]
        acc[k] += 1
        kinds.append(k)
    order = rng.permutation(nseg)
    kinds = [kinds[i] for i in order]
    # generate each kind once at its total size, then cut segments (keeps generators vectorised)
    need = {k: kinds.count(k) * segment for k in set(kinds)}
    pools = {k: make(k, need[k], seed + 17 * i) for i, k in enumerate(sorted(need))}
    used = {k: 0 for k in need}
    parts = []
    for k in kinds:
        parts.append(pools[k][used[k]:used[k] + segment])
        used[k] += segment
    return np.ascontiguousarray(np.concatenate(parts)[:nbytes])
