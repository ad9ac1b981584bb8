#!/usr/bin/env python3
"""Experiment (CPU): how the Zstd chain batches resolve on the bench corpus, from the lane model in tests/."""
import ctypes as C, numpy as np, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aocl_compression_amd import corpus
L = C.CDLL(os.path.join(ROOT, "tests", "_build", "libzchain_model.so")); L.zchain_model_sequences.restype = C.c_size_t
L.zchain_model_sequences.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p, C.c_int]
kinds = sys.argv[1:] or ["mix"]
for kind in kinds:
    data = (corpus.silesia_mix(16 << 20) if kind == "mix" else corpus.make(kind, 4 << 20)).tobytes()
    st = np.zeros(40, dtype=np.uint64); tot = 0
    for i in range(0, len(data), 131072):
        d = data[i:i + 131072]; out = np.zeros(4 * (len(d) // 3 + 16), dtype=np.uint32); lit = C.c_size_t(0); s = np.zeros(40, dtype=np.uint64)
        tot += L.zchain_model_sequences(d, len(d), out.ctypes.data, C.byref(lit), s.ctypes.data, 2); st += s
    b, f, sp, rs, co, sq = [int(x) for x in st[:6]]
    print(f"{kind:9s} seqs {tot} batches {b} fast {f} ({f/max(tot,1):.2f}) special {sp} restart {rs} cont {co} seqsearch {sq} seq/batch {tot/max(b,1):.2f}")
    print("   special reasons (1 long ext, 2 catch-up, 4 writes beyond the batch, 16 rep2 hit):", {k: int(st[8 + k]) for k in range(32) if st[8 + k]})
