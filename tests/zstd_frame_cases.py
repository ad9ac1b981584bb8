"""Inputs for the multi-block Zstd frame tests (inputs above 128 KiB become one frame of 128 KiB blocks that share the hash
table, the repeat offsets and the Huffman table: zstd_compress.c:4504-4583).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import random

import numpy as np

import aocl_compression_amd as a


class CParams(C.Structure):
    _fields_ = [(k, C.c_uint) for k in ("windowLog", "chainLog", "hashLog", "searchLog", "minMatch", "targetLength")] + [("strategy", C.c_int)]


class FParams(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("contentSizeFlag", "checksumFlag", "noDictIDFlag")]


class ZParams(C.Structure):
    _fields_ = [("c", CParams), ("f", FParams)]


def piece(kind, n, rng):
    seed = rng.randint(0, 1 << 30)
    if kind == "zeros":
        return bytes([rng.randint(0, 255)]) * n
    if kind == "low":                                   # 2 bits of entropy per byte: long literal runs the Huffman stage shrinks
        return np.random.default_rng(seed).integers(0, 4, n, dtype=np.uint8).tobytes()
    if kind == "rep":                                   # a short period: almost no literals after the first block
        p = np.random.default_rng(seed).integers(0, 256, rng.randint(1, 5000), dtype=np.uint8).tobytes()
        return (p * (n // len(p) + 1))[:n]
    return a.corpus.make(kind, n, seed=seed).tobytes()


def make(kind, n, seed):
    """`kind`: a corpus kind, "low", "rep", "zeros" or "mixed" (pieces of all of them, 1 KB - 300 KB each, so that blocks
    change statistics, end raw, turn into RLE blocks and reuse or replace the previous block's Huffman table)."""
    rng = random.Random(seed)
    if kind != "mixed":
        return piece(kind, n, rng)
    out = bytearray()
    kinds = list(a.corpus.KINDS) + ["low", "rep", "zeros", "random"]
    while len(out) < n:
        out += piece(rng.choice(kinds), rng.randint(1000, 300000), rng)
    return bytes(out[:n])


KINDS = ("text", "mixed", "low", "rep", "zeros", "random", "jsonlog")
EDGE_SIZES = (131073, 131078, 131079, 200000, 262144, 262145, 300000, 524288, 524289, (1 << 20) + 5)


def ref_frame(ref, data, cp, extra=0):
    """the reference's ZSTD_compress_advanced with these compression parameters (contentSizeFlag 1, no checksum)"""
    L = ref.L
    L.ZSTD_createCCtx.restype = C.c_void_p
    L.ZSTD_freeCCtx.argtypes = [C.c_void_p]
    L.ZSTD_compress_advanced.restype = C.c_size_t
    L.ZSTD_compress_advanced.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, ZParams]
    n = len(data)
    cap = n + (n >> 8) + 64 + extra
    out = C.create_string_buffer(cap)
    cctx = L.ZSTD_createCCtx()
    try:
        r = L.ZSTD_compress_advanced(cctx, out, cap, data, n, None, 0, ZParams(cp, FParams(1, 0, 0)))
    finally:
        L.ZSTD_freeCCtx(cctx)
    assert r < (1 << 62), "reference error"
    return out.raw[:r]


def ref_level_cparams(ref, level, n):
    L = ref.L
    L.aocl_setup_zstd_encode.restype = C.c_void_p
    L.aocl_setup_zstd_encode.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t]
    L.aocl_setup_zstd_encode(0, level, 0, 3, 0)          # the dispatcher selects the AOCL level table (zstd_compress.c:7390-7425)
    L.ZSTD_getCParams.restype = CParams
    L.ZSTD_getCParams.argtypes = [C.c_int, C.c_ulonglong, C.c_size_t]
    return L.ZSTD_getCParams(level, n, 0)


def count164_case(seed, ties=12):
    """bytes whose histogram has `ties` symbols with a count of exactly 164 (shuffled, so almost no matches): the reference's
    Huffman sort quicksorts that bucket too (its loop starts one bucket below the log2 buckets, huf_compress.c:649-664) and
    permutes nine or more equal counts - which symbol of them gets the shorter code is part of the output"""
    rng = np.random.default_rng(seed)
    counts = rng.integers(100, 900, 256)
    counts[rng.choice(256, ties, replace=False)] = 164
    counts[counts == 164] = 164                          # (others that drew 164 by chance simply join the bucket)
    return rng.permutation(np.repeat(np.arange(256, dtype=np.uint8), counts)).tobytes()
