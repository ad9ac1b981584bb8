#!/usr/bin/env python3
"""Driver of tools/experiments/lz4_segment_study.c (CPU only): per corpus kind, how often a cold-started byU32 LZ4 parser is in
the true parser's state behind a segment boundary.  Usage: lz4_segment_study.py [MiB per kind]"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import aocl_compression_amd as a

so = "/tmp/liblz4study.so"
subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "experiments", "lz4_segment_study.c")], check=True)
L = C.CDLL(so)
L.lz4_segment_study.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int]
mib = float(sys.argv[1]) if len(sys.argv) > 1 else 16
n = int(mib * (1 << 20))
kinds = ["text", "jsonlog", "binary", "database", "image16", "mix"]
print(f"{mib:g} MiB per kind; segment 256 KiB, zone 4 KiB")
print(f"{'kind':9s} {'warm':>8s} {'bound.':>6s} {'sync':>6s} {'strict':>6s} {'actual':>6s} {'sync dist (median/p95/max)':>28s} {'strict dist (median/p95)':>24s} {'diff slots at sync (mean)':>24s}")
for kind in kinds:
    data = a.corpus.silesia_mix(n, seed=0xA0C1) if kind == "mix" else a.corpus.make(kind, n, seed=0xA0C1)
    data = np.ascontiguousarray(data)
    for warm in (16 << 10, 64 << 10, 128 << 10, 256 << 10):
        out = np.zeros(8 * 4096, np.uint32)
        nb = L.lz4_segment_study(data.ctypes.data, n, 256 << 10, warm, 4096, out.ctypes.data, 4096)
        r = out[:8 * nb].reshape(nb, 8)
        sync = r[:, 1] > 0; strict = r[:, 2] > 0; actual = r[:, 3] > 0
        sd = (r[sync, 1] - r[sync, 0]) if sync.any() else np.array([0])
        td = (r[strict, 2] - r[strict, 0]) if strict.any() else np.array([0])
        print(f"{kind:9s} {warm >> 10:6d}Ki {nb:6d} {sync.mean() * 100:5.1f}% {strict.mean() * 100:5.1f}% {actual.mean() * 100:5.1f}% "
              f"{int(np.median(sd)):10d}/{int(np.percentile(sd, 95)):6d}/{int(sd.max()):6d} {int(np.median(td)):12d}/{int(np.percentile(td, 95)):6d} {r[sync, 4].mean() if sync.any() else 0:20.1f}")
