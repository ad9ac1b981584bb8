"""CPU check of the Zstd match finder's batch/chain logic: tests/zstd_chain_model.cpp (a lane-by-lane model of the
kernel's batch that runs the kernel's own scalar walk, csrc/zstd_chain.h) must produce exactly the oracle's sequences."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
from aocl_compression_amd import corpus

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO = os.path.join(HERE, "_build", "libzchain_model.so")


@pytest.fixture(scope="module")
def model():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-Wall", "-I", os.path.join(ROOT, "aocl-compression_amd", "csrc"),
                    os.path.join(HERE, "zstd_chain_model.cpp"), "-o", SO], check=True)
    L = C.CDLL(SO)
    L.zchain_model_sequences.restype = C.c_size_t
    L.zchain_model_sequences.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p, C.c_int]
    return L


@pytest.fixture(scope="module")
def orc():
    ol.build_oracle()
    L = C.CDLL(ol.ORACLE_SO)
    L.orc_dbg_zstd_sequences.restype = C.c_size_t
    L.orc_dbg_zstd_sequences.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t)]
    return L


def sequences(fn, data: bytes, *extra):
    n = len(data)
    out = np.zeros(4 * (n // 3 + 16), dtype=np.uint32)
    lit = C.c_size_t(0)
    buf = C.create_string_buffer(data, n + 1)
    nb = fn(buf, n, out.ctypes.data, C.byref(lit), *extra)
    return out[:4 * nb].reshape(-1, 4)[:, :3].copy(), lit.value


def check(model, orc, data: bytes, stats_acc=None):
    want, wl = sequences(orc.orc_dbg_zstd_sequences, data)
    st = np.zeros(8, dtype=np.uint64)
    got, gl = sequences(model.zchain_model_sequences, data, st.ctypes.data, 1)
    if stats_acc is not None:
        stats_acc += st
    if got.shape != want.shape or not np.array_equal(got, want) or gl != wl:
        k = 0
        while k < min(len(got), len(want)) and np.array_equal(got[k], want[k]):
            k += 1
        raise AssertionError(f"first difference at sequence {k}: model {got[k:k+3].tolist()} oracle {want[k:k+3].tolist()} (n={len(data)})")
    seq0, l0 = sequences(model.zchain_model_sequences, data, None, 0)          # the model's sequential path alone
    assert np.array_equal(seq0, want) and l0 == wl
    # searches resumed inside the batch they start in (4), only when a matching lane is left (8), long matches that end inside
    # the batch taken from the equality map of their own distance (16)
    # (+32: duplicate lanes do not cut the batch - the walk stops at them and they are probed again: measured on the GPU in round 4,
    #  slower than the cut, not in the kernel; the rule itself is pinned here)
    for mode in (5 | (8 << 8), 5 | (1 << 8), 29 | (8 << 8), 29 | (2 << 8), 1 | 32, 5 | 32 | (8 << 8)):
        seq5, l5 = sequences(model.zchain_model_sequences, data, None, mode)
        assert np.array_equal(seq5, want) and l5 == wl, f"resumed batches (mode {mode:#x}) differ from the oracle"


@pytest.mark.parametrize("kind", sorted(corpus.KINDS))
def test_corpus_kinds(model, orc, kind):
    data = corpus.make(kind, 3 << 20).tobytes()
    acc = np.zeros(8, dtype=np.uint64)
    for i in range(0, len(data), 131072):
        check(model, orc, data[i:i + 131072], acc)
    assert acc[1] > 0 or kind == "random"            # the chain path is exercised


def test_mix_and_sizes(model, orc):
    data = corpus.silesia_mix(8 << 20).tobytes()
    for i in range(0, len(data), 131072):
        check(model, orc, data[i:i + 131072])
    for n in [0, 1, 19, 20, 21, 99, 119, 120, 121, 130, 200, 1000, 4095, 16384, 16385, 65536, 100000, 131071]:
        check(model, orc, data[12345:12345 + n])


def lz_grammar(rng, nbytes, alphabet):
    out = bytearray()
    while len(out) < nbytes:
        r = rng.integers(0, 10)
        if r < 4 or not out:
            out += bytes(rng.integers(0, alphabet, int(rng.integers(1, 24)), dtype=np.uint8))
        elif r < 8:
            d = int(rng.integers(1, min(len(out), 70000) + 1)); ln = int(rng.integers(3, 60))
            for _ in range(ln):
                out.append(out[-d])
        elif r < 9:
            out += bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 80))
        else:
            p = int(rng.integers(1, 10))
            if len(out) >= p:
                for _ in range(int(rng.integers(4, 120))):
                    out.append(out[-p])
    return bytes(out[:nbytes])


def test_zero_bytes8_matches_the_plain_form(model):
    """csrc/zstd_chain.h zero_bytes8: the eight zero-byte flags of a dword pair gathered by one multiplication"""
    model.zchain_model_zero_bytes8_check.restype = C.c_size_t
    model.zchain_model_zero_bytes8_check.argtypes = [C.c_size_t, C.c_uint64]
    assert model.zchain_model_zero_bytes8_check(20_000_000, 0x9E3779B97F4A7C15) == 0


def test_search_that_starts_at_the_last_lane(model, orc):
    """tests/golden/zstd_walk_search_at_lane63.bin (found by tools/fuzz_campaign.py, seed 9903): a repcode match ends at lane 62
    of a batch, so the next search starts at lane 63 - its repcode test needs bytes behind the batch and the walk has to stop
    there instead of taking lane 63's table match (the shift that builds the repcode events of the search was by 64)."""
    check(model, orc, open(os.path.join(HERE, "golden", "zstd_walk_search_at_lane63.bin"), "rb").read())


@pytest.mark.parametrize("alphabet", [2, 4, 16, 256])
def test_grammar_inputs(model, orc, alphabet):
    rng = np.random.default_rng(1234 + alphabet)
    for it in range(40):
        n = int(rng.integers(100, 60000))
        check(model, orc, lz_grammar(rng, n, alphabet))


def test_record_like_inputs(model, orc):
    """fixed-size records with a few changing bytes: repcode after repcode, immediate rep2 matches"""
    rng = np.random.default_rng(7)
    for rec in [8, 12, 16, 24, 37, 64]:
        base = rng.integers(0, 256, rec, dtype=np.uint8)
        rows = np.tile(base, (131072 // rec, 1))
        for k in range(rows.shape[0]):
            for _ in range(int(rng.integers(0, 3))):
                rows[k, int(rng.integers(0, rec))] = int(rng.integers(0, 256))
        check(model, orc, rows.tobytes())


# ---- frames: blocks that continue from the table, window and repeat offsets of the block before (fast_block<.., FRAME>) ----
def frame_rows(fn, data: bytes, wl, hl, mm, *extra):
    n = len(data)
    out = np.zeros(4 * (n // 3 + n // 1024 + 64), dtype=np.uint32)
    buf = C.create_string_buffer(data, n + 1)
    rows = fn(buf, n, wl, hl, mm, out.ctypes.data, *extra)
    return out[:4 * rows].reshape(-1, 4)[:, :3].copy()


def check_frame(model, orc, data: bytes, wl, hl, mm):
    model.zchain_model_frame_sequences.restype = C.c_size_t
    model.zchain_model_frame_sequences.argtypes = [C.c_void_p, C.c_size_t, C.c_uint, C.c_uint, C.c_uint, C.c_void_p, C.c_int]
    orc.orc_dbg_zstd_frame_sequences.restype = C.c_size_t
    orc.orc_dbg_zstd_frame_sequences.argtypes = [C.c_void_p, C.c_size_t, C.c_uint, C.c_uint, C.c_uint, C.c_void_p]
    want = frame_rows(orc.orc_dbg_zstd_frame_sequences, data, wl, hl, mm)
    for use_chain in (1, 0):
        got = frame_rows(model.zchain_model_frame_sequences, data, wl, hl, mm, use_chain)
        if got.shape != want.shape or not np.array_equal(got, want):
            k = 0
            while k < min(len(got), len(want)) and np.array_equal(got[k], want[k]):
                k += 1
            raise AssertionError(f"frame (wl {wl}, hl {hl}, mm {mm}, chain {use_chain}): first difference at row {k}: "
                                 f"model {got[k:k+3].tolist()} oracle {want[k:k+3].tolist()} (n={len(data)})")


@pytest.mark.parametrize("kind", ["text", "mixed", "low", "rep", "jsonlog"])
def test_frames(model, orc, kind):
    """the lane model of the frame instantiation of the finder against the oracle's block-after-block parser: windows above the
    input, below it, equal to the block size (the prefix start moves, repeat offsets reach below it), every minMatch"""
    import zstd_frame_cases as zfc
    for i, n in enumerate((131073, 200000, 300000, 524289, 1000000)):
        data = zfc.make(kind, n, seed=900 + i)
        for (wl, hl, mm) in ((20, 14, 7), (18, 14, 6), (17, 13, 5), (12, 11, 4), (10, 11, 7), (15, 9, 6)):
            check_frame(model, orc, data, wl, hl, mm)


def test_frames_grammar(model, orc):
    rng = np.random.default_rng(99)
    for it in range(30):
        n = int(rng.integers(1000, 400000))
        data = lz_grammar(rng, n, int(rng.choice([2, 4, 16, 256])))
        check_frame(model, orc, data, int(rng.integers(10, 21)), int(rng.integers(6, 15)), int(rng.integers(4, 8)))
