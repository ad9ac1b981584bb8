"""The LZ4 decoder's group bookkeeping on the CPU (tests/lz4_group_model.py, a lane-by-lane model of lz4_decompress_wave's scan /
record / group-execution logic) against the oracle: every corpus kind, ragged sizes, long literal runs and long matches, inputs
whose groups overflow the 2 KB block (records carried over) and the fuzz input that exposed the refill-after-rewind bug."""
import os

import numpy as np
import pytest

import aocl_compression_amd as a
import lz4_group_model as model
import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def orc():
    return ol.Oracle()


@pytest.mark.parametrize("kind", list(a.corpus.KINDS))
def test_model_decodes_what_the_oracle_compresses(orc, kind):
    raw = a.corpus.make(kind, 40000, seed=11).tobytes()
    tr = {}
    assert model.decode(orc.compress(ol.LZ4, raw), len(raw), tr) == raw
    assert kind == "random" or (tr["groups"] > 0 and tr["seq"] > 0)          # (random bytes: one run of literals, the one-sequence path)


@pytest.mark.parametrize("n", [0, 1, 12, 13, 64, 351, 352, 353, 1000, 4096, 70000])
def test_model_ragged_sizes(orc, n):
    raw = a.corpus.make("text", n, seed=3).tobytes()
    assert model.decode(orc.compress(ol.LZ4, raw), max(n, 0)) == raw


def _lz_text(rng, nseq, lit_hi, mlo, mhi):
    out = bytearray(rng.integers(0, 256, 400, dtype=np.uint8).tobytes())
    for _ in range(nseq):
        out += rng.integers(0, 256, int(rng.integers(1, lit_hi)), dtype=np.uint8).tobytes()
        d = int(rng.integers(300, min(len(out), 60000) + 1)) if len(out) > 300 else len(out)
        n = int(rng.integers(mlo, mhi))
        seg = bytes(out[-d:])
        out += (seg * (n // len(seg) + 1))[:n]
    return bytes(out)


def test_model_long_fields(orc):
    raw = _lz_text(np.random.default_rng(5), 1500, 600, 4, 600)[:200000]      # runs and matches of up to 600 bytes
    tr = {}
    assert model.decode(orc.compress(ol.LZ4, raw), len(raw), tr) == raw
    assert tr["steps"] > 1 and tr["groups"] > 0, tr            # sequences of the one-at-a-time path between groups


def test_model_groups_that_overflow_the_block(orc):
    raw = _lz_text(np.random.default_rng(6), 3000, 20, 40, 200)[:250000]       # ~130 output bytes per sequence: 64 of them do not fit 2 KB
    tr = {}
    assert model.decode(orc.compress(ol.LZ4, raw), len(raw), tr) == raw
    assert tr["carried"] > 0, tr


def test_model_fuzz_input_that_rewinds(orc):
    raw = open(os.path.join(ROOT, "tests", "golden", "lz4_decode_rewind_seed13759_chunk2.bin"), "rb").read()
    assert model.decode(orc.compress(ol.LZ4, raw), len(raw)) == raw


def test_model_rejects_bad_offsets(orc):
    raw = a.corpus.make("text", 5000, seed=1).tobytes()
    comp = bytearray(orc.compress(ol.LZ4, raw))
    # the first sequence's offset reaches before the start of the output
    lit = comp[0] >> 4; pos = 1
    if lit == 15:
        while True:
            b = comp[pos]; pos += 1; lit += b
            if b != 255: break
    comp[pos + lit] = 0xFF; comp[pos + lit + 1] = 0xFF
    with pytest.raises(model.Malformed):
        model.decode(bytes(comp), len(raw))
    assert orc.decompress(ol.LZ4, bytes(comp), len(raw)) is None
