"""The LZ4 compressor's batch logic on the CPU (tests/lz4_batch_model.py: 64 consecutive positions read the table as the batch
found it, duplicate lanes are decided when the walk arrives, the latest writer owns a slot) against the oracle's parse: the
sequences of every block must be the reference's - on every corpus kind, on inputs full of repeats at distances below 64 (the
duplicate lanes), on runs, and on ragged sizes."""
import numpy as np
import pytest

import aocl_compression_amd as a
import lz4_batch_model as model
import oracle_lib as ol


@pytest.fixture(scope="module")
def orc():
    return ol.Oracle()


def check(orc, raw: bytes):
    want = model.parse_sequences(orc.compress(ol.LZ4, raw)) if len(raw) else []
    m = model.Model(raw)
    got = m.run()
    if got != want:
        k = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), min(len(got), len(want)))
        raise AssertionError(f"sequence {k}: model {got[k:k + 2]} reference {want[k:k + 2]} (n={len(raw)}, {len(got)} vs {len(want)} sequences)")
    return m.stats


@pytest.mark.parametrize("kind", list(a.corpus.KINDS))
def test_corpus_kinds(orc, kind):
    st = check(orc, a.corpus.make(kind, 150000, seed=5).tobytes())
    assert st["batches"] > 0


@pytest.mark.parametrize("size", [60000, 90000])          # the 16-bit table of inputs below 64 KB and the 32-bit one
def test_short_distance_repeats_and_runs(orc, size):
    """what makes duplicate lanes: periodic records, repeats 5 - 60 bytes back, runs of one byte, all mixed with noise"""
    rng = np.random.default_rng(size)
    tot = dict(dup_stops=0, dup_inbatch=0, dup_table=0, dup_nohit=0)
    for it in range(8):
        out = bytearray(rng.integers(0, 256, 300, dtype=np.uint8).tobytes())
        while len(out) < size:
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
                d = int(rng.integers(64, min(len(out), 65000))); ln = int(rng.integers(4, 80))
                for _ in range(ln): out.append(out[-d])
            else:
                rec = bytearray(out[-24:]); rec[int(rng.integers(0, 24))] ^= 0x55; out += rec
        st = check(orc, bytes(out[:size]))
        for k in tot: tot[k] += st[k]
    assert tot["dup_inbatch"] > 1000 and tot["dup_table"] > 0 and tot["dup_nohit"] > 0


@pytest.mark.parametrize("n", [0, 1, 12, 13, 14, 64, 100, 104, 105, 106, 200, 1000, 4097, 65536, 70000])
def test_ragged_sizes(orc, n):
    check(orc, a.corpus.make("binary", n, seed=9).tobytes() if n else b"")
