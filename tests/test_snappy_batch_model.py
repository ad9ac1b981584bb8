"""The Snappy compressor's batch rule on the CPU (tests/snappy_batch_model.py: in the batch of consecutive positions that follows
a copy, a duplicate lane is decided when the walk arrives - latest earlier written lane of its slot, else the table's entry)
against the copies of the oracle's streams, fragment by fragment: every corpus kind, inputs made of repeats below 64 bytes and
runs, ragged sizes."""
import numpy as np
import pytest

import aocl_compression_amd as a
import oracle_lib as ol
import snappy_batch_model as model


@pytest.fixture(scope="module")
def orc():
    return ol.Oracle()


def check(orc, raw: bytes):
    tot = {}
    for p in range(0, len(raw), model.BLOCK):
        frag = raw[p:p + model.BLOCK]
        n, want = model.parse_elements(orc.compress(ol.SNAPPY, frag))
        assert n == len(frag)
        f = model.Fragment(frag)
        got = f.run()
        if got != want:
            k = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), min(len(got), len(want)))
            raise AssertionError(f"fragment at {p}, copy {k}: model {got[k:k + 2]} reference {want[k:k + 2]} ({len(got)} vs {len(want)} copies)")
        for k, v in f.stats.items(): tot[k] = tot.get(k, 0) + v
    return tot


@pytest.mark.parametrize("kind", list(a.corpus.KINDS))
def test_corpus_kinds(orc, kind):
    st = check(orc, a.corpus.make(kind, 200000, seed=6).tobytes())
    assert st["batches"] > 0 or kind == "random"              # (noise holds no copies)


def test_short_distance_repeats_and_runs(orc):
    rng = np.random.default_rng(11)
    tot = dict(dup_inbatch=0, dup_table=0, dup_nohit=0)
    for it in range(6):
        out = bytearray(rng.integers(0, 256, 300, dtype=np.uint8).tobytes())
        while len(out) < 100000:
            r = int(rng.integers(0, 6))
            if r == 0: out += rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8).tobytes()
            elif r == 1: out += bytes([int(rng.integers(0, 256))]) * int(rng.integers(2, 90))
            elif r == 2:
                p = int(rng.integers(2, 61)); base = bytes(out[-p:])
                out += (base * 8)[:int(rng.integers(4, 200))]
            elif r == 3:
                d = int(rng.integers(5, 64)); ln = int(rng.integers(4, 30))
                for _ in range(ln): out.append(out[-d])
            elif r == 4:
                d = int(rng.integers(64, min(len(out), 60000))); ln = int(rng.integers(4, 80))
                for _ in range(ln): out.append(out[-d])
            else:
                rec = bytearray(out[-24:]); rec[int(rng.integers(0, 24))] ^= 0x55; out += rec
        st = check(orc, bytes(out[:100000]))
        for k in tot: tot[k] += st[k]
    assert tot["dup_inbatch"] > 1000 and tot["dup_table"] > 0 and tot["dup_nohit"] > 0


@pytest.mark.parametrize("n", [1, 14, 15, 16, 64, 100, 119, 120, 121, 200, 1000, 4097, 65535, 65536, 65537, 70000])
def test_ragged_sizes(orc, n):
    check(orc, a.corpus.make("binary", n, seed=9).tobytes())
