"""CPU check of the speculative Huffman literal decoder's per-lane code (csrc/zstd_hufspec.h) through
tests/zstd_hufspec_model.cpp: 16 lanes run one after the other with the kernel's repair rounds, and must reproduce the plain
sequential decoder - symbols and accept / reject - on random prefix codes (table logs 1-11), on skewed and flat symbol
statistics, on tiny streams, and on damaged streams (flipped bits, wrong symbol counts, cut tails)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO = os.path.join(HERE, "_build", "libhufspec_model.so")


@pytest.fixture(scope="module")
def model():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-Wall", "-I", os.path.join(ROOT, "aocl-compression_amd", "csrc"),
                    os.path.join(HERE, "zstd_hufspec_model.cpp"), "-o", SO], check=True)
    L = C.CDLL(SO)
    for f in (L.hufspec_seq_decode, L.hufspec_model_decode):
        f.restype = C.c_int
    L.hufspec_seq_decode.argtypes = [C.c_char_p, C.c_uint, C.c_void_p, C.c_uint, C.c_uint, C.c_void_p]
    L.hufspec_model_decode.argtypes = [C.c_char_p, C.c_uint, C.c_void_p, C.c_uint, C.c_uint, C.c_void_p, C.c_int, C.c_void_p]
    return L


def random_code(rng, nsym, tl):
    """code lengths of a complete prefix code with nsym leaves, none longer than tl (leaves are split at random)"""
    leaves = [0]
    while len(leaves) < nsym:
        cand = [i for i, l in enumerate(leaves) if l < tl]
        if not cand:
            break
        i = cand[int(rng.integers(len(cand)))] if rng.random() < 0.5 else min(cand, key=lambda k: leaves[k])
        l = leaves.pop(i)
        leaves += [l + 1, l + 1]
    return leaves


def build(lens_by_symbol, tl):
    """X1 cells (symbol | nbBits << 8) in the order of HUF_readDTableX1 (weight ascending, symbols ascending) and the code
    value of every symbol"""
    cells = np.zeros(1 << tl, dtype=np.uint16)
    codes = {}
    pos = 0
    for w in range(1, tl + 1):
        for s, l in sorted(lens_by_symbol.items()):
            if tl + 1 - l == w:
                n = 1 << (w - 1)
                cells[pos:pos + n] = s | (l << 8)
                codes[s] = (pos >> (tl - l), l)
                pos += n
    assert pos == 1 << tl
    return cells, codes


def encode(symbols, codes):
    """backward bitstream: the first symbol's code sits right below the end mark, most significant bit first"""
    total = sum(codes[s][1] for s in symbols)
    bits = np.zeros(total + 1, dtype=np.uint8)
    p = total
    for s in symbols:
        v, l = codes[s]
        for k in range(l):
            bits[p - 1 - k] = (v >> (l - 1 - k)) & 1
        p -= l
    bits[total] = 1
    pad = (-len(bits)) % 8
    return np.packbits(np.concatenate([bits, np.zeros(pad, dtype=np.uint8)]), bitorder="little").tobytes()


def both(model, stream, cells, tl, nout, lanes=16):
    a = np.zeros(nout + 8, dtype=np.uint8)
    b = np.zeros(nout + 8, dtype=np.uint8)
    st = np.zeros(2, dtype=np.uint32)
    r0 = model.hufspec_seq_decode(stream, len(stream), cells.ctypes.data, tl, nout, a.ctypes.data)
    r1 = model.hufspec_model_decode(stream, len(stream), cells.ctypes.data, tl, nout, b.ctypes.data, lanes, st.ctypes.data)
    return r0, a[:nout], r1, b[:nout], st


def make_case(rng, tl, nsym, nout, skew):
    lens = random_code(rng, nsym, tl)
    tl_eff = max(lens) if max(lens) > 0 else 1
    syms = rng.permutation(256)[:len(lens)]
    by = {int(s): int(l) for s, l in zip(syms, lens)}
    cells, codes = build(by, tl_eff)
    p = np.array([2.0 ** (-l * skew) for l in lens]); p /= p.sum()
    data = [int(x) for x in rng.choice(syms, size=nout, p=p)]
    return encode(data, codes), cells, tl_eff, data


@pytest.mark.parametrize("tl", [1, 2, 5, 8, 9, 10, 11])
def test_random_codes(model, tl):
    rng = np.random.default_rng(100 + tl)
    worst = 0
    for it in range(40):
        nsym = int(rng.integers(2, min(256, 1 << tl) + 1))
        nout = int(rng.choice([1, 2, 3, 7, 16, 63, 64, 257, 1000, 4097, 20000]))
        skew = float(rng.choice([0.0, 0.5, 1.0, 1.5]))
        stream, cells, tle, data = make_case(rng, tl, nsym, nout, skew)
        r0, a, r1, b, st = both(model, stream, cells, tle, nout)
        assert r0 == 0 and r1 == 0, (tl, it, r0, r1)
        assert np.array_equal(a, np.array(data, dtype=np.uint8)) and np.array_equal(a, b), (tl, it)
        worst = max(worst, int(st[0]))
    assert worst <= 18                                # (a probe round in front of the rounds that settle lane after lane)


def test_other_lane_counts(model):
    rng = np.random.default_rng(7)
    for lanes in (1, 2, 3, 16, 64):
        stream, cells, tle, data = make_case(rng, 11, 200, 30000, 1.0)
        r0, a, r1, b, st = both(model, stream, cells, tle, len(data), lanes)
        assert r0 == 0 and r1 == 0 and np.array_equal(a, b)


def test_work_is_a_fraction_of_a_sequential_pass(model):
    rng = np.random.default_rng(11)
    stream, cells, tle, data = make_case(rng, 11, 120, 32768, 1.0)
    r0, a, r1, b, st = both(model, stream, cells, tle, len(data))
    assert r0 == 0 and r1 == 0 and np.array_equal(a, b)
    assert st[0] <= 4 and st[1] < 2.6 * len(data)       # a probe of the segments' tails, one counting run, the writing run (three full passes before round 6)


def test_damaged_streams_agree_with_the_sequential_decoder(model):
    rng = np.random.default_rng(23)
    agree = rejected = 0
    for it in range(300):
        tl = int(rng.integers(3, 12))
        nout = int(rng.choice([5, 64, 300, 2000, 9000]))
        stream, cells, tle, data = make_case(rng, tl, int(rng.integers(2, 200)), nout, 1.0)
        s = bytearray(stream)
        kind = it % 4
        claim = nout
        if kind == 0:
            s[int(rng.integers(len(s)))] ^= 1 << int(rng.integers(8))
        elif kind == 1:
            claim = max(1, nout + int(rng.integers(-3, 4)))
        elif kind == 2:
            s = s[:max(1, len(s) - int(rng.integers(1, 4)))]
        else:
            s = bytearray(rng.integers(0, 256, size=max(2, len(s) // 2), dtype=np.uint8).tobytes())
        if s[-1] == 0:
            s[-1] = 1
        r0, a, r1, b, st = both(model, bytes(s), cells, tle, claim)
        assert (r0 == 0) == (r1 == 0), (it, kind, r0, r1)
        if r0 == 0:
            assert np.array_equal(a, b)
            agree += 1
        else:
            rejected += 1
    assert rejected > 100 and agree > 10
