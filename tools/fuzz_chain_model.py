#!/usr/bin/env python3
"""CPU fuzz of the Zstd match finder's batch / walk logic (TEST INFRASTRUCTURE: loads oracle/): the lane-by-lane model of
tests/zstd_chain_model.cpp, which runs the kernel's own csrc/zstd_chain.h, against the oracle's sequential parser on the input
families of tools/fuzz_campaign.py.  No GPU needed.  usage: tools/fuzz_chain_model.py <seconds> [seed]"""
import ctypes as C, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
import test_zstd_chain_model as t

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
so = os.path.join(ROOT, "tests", "_build", f"libzchain_model_fuzz_{os.getpid()}.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-I", os.path.join(ROOT, "aocl-compression_amd", "csrc"), os.path.join(ROOT, "tests", "zstd_chain_model.cpp"), "-o", so], check=True)
M = C.CDLL(so)
M.zchain_model_sequences.restype = C.c_size_t
M.zchain_model_sequences.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p, C.c_int]
ol.build_oracle()
O = C.CDLL(ol.ORACLE_SO)
O.orc_dbg_zstd_sequences.restype = C.c_size_t
O.orc_dbg_zstd_sequences.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t)]


def build(rng, target):
    out = bytearray()
    alphabet = int(rng.choice([2, 4, 16, 256]))
    while len(out) < target:
        k = rng.integers(0, 5)
        if k == 0:
            out += bytes(rng.integers(0, alphabet, int(rng.integers(1, 60)), dtype=np.uint8))
        elif k == 1 and out:
            d = 1 + int(rng.integers(0, min(len(out), 70000))); n = int(rng.integers(3, 400))
            if d >= n: out += out[len(out) - d:len(out) - d + n]
            else: out += (bytes(out[-d:]) * (n // d + 1))[:n]
        elif k == 2:
            out += bytes([int(rng.integers(0, alphabet))]) * int(rng.integers(1, 300))
        elif k == 3 and len(out) > 8:
            p = int(rng.integers(1, 9)); out += (bytes(out[-p:]) * 40)[:int(rng.integers(4, 200))]
        else:
            out += bytes(rng.integers(0, 256, int(rng.integers(1, 12)), dtype=np.uint8))
    return bytes(out[:target])


def seq_shaped(rng, n):
    lls = rng.choice([0, 1, 2, 3, 5, 14, 15, 16, 17, 59, 60, 61, 270], int(rng.integers(1, 5)))
    mls = rng.choice([4, 5, 6, 7, 8, 11, 12, 18, 19, 20, 27, 28, 29, 34, 63, 64, 65, 130, 259, 1000], int(rng.integers(1, 5)))
    ofs = rng.choice([1, 2, 3, 4, 7, 8, 9, 16, 31, 32, 33, 100, 255, 256, 769, 2048, 4096, 20000, 65535, 65536], int(rng.integers(1, 5)))
    out = bytearray(rng.integers(0, 256, int(rng.integers(8, 3000)), dtype=np.uint8).tobytes())
    while len(out) < n:
        out += rng.integers(0, 256, int(rng.choice(lls)), dtype=np.uint8).tobytes()
        d = min(int(rng.choice(ofs)), len(out)); m = int(rng.choice(mls))
        if d >= m: out += out[len(out) - d:len(out) - d + m]
        else: out += (bytes(out[-d:]) * (m // d + 1))[:m]
    return bytes(out[:n])


t_end = time.time() + budget
it = nbytes = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed0 + it); it += 1
    size = int(rng.choice([300, 5000, 17000, 40000, 65547, 100000, 131072]))
    raw = seq_shaped(rng, size) if rng.random() < 0.5 else build(rng, size)
    try:
        t.check(M, O, raw)
    except AssertionError as e:
        print(f"FAIL seed {seed0 + it - 1} size {size}: {e}"); sys.exit(1)
    nbytes += size
os.remove(so)
print(f"chain model fuzz OK: {it} inputs, {nbytes / 1e6:.1f} MB, seeds {seed0}..{seed0 + it - 1}")
