#!/usr/bin/env python3
"""Compress / decompress rate of one codec per corpus kind (development tool): which data the kernels are slow on.
usage: tools/kind_bench.py [lz4|snappy|zstd] [MiB per kind]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aocl_compression_amd as m
codec = {"lz4": (m.LZ4, 262144), "snappy": (m.SNAPPY, 65536), "zstd": (m.ZSTD, 131072)}[sys.argv[1] if len(sys.argv) > 1 else "lz4"]
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
eng = m.engine(0)
for kind in m.corpus.KINDS:
    base = torch.from_numpy(m.corpus.make(kind, 64 << 20, seed=7)).to(eng.device)
    total = (mib << 20) // codec[1] * codec[1]
    src = base.repeat(-(-total // base.numel()))[:total].contiguous()
    n = total // codec[1]
    slot = (m.compress_bound(codec[0], codec[1]) + 15) & ~15
    slots = torch.empty(n * slot, dtype=torch.uint8, device=eng.device); clen = torch.zeros(n, dtype=torch.int32, device=eng.device)
    dst = torch.empty(total, dtype=torch.uint8, device=eng.device); dlen = torch.zeros(n, dtype=torch.int32, device=eng.device)
    cb = eng.uniform_compress_batch(src, codec[1], slots, clen, slot)
    db = eng.uniform_decompress_batch(slots, clen, slot, dst, codec[1], dlen)
    eng.time_batch(codec[0], 0, cb, 1); tc = min(eng.time_batch(codec[0], 0, cb, 1) for _ in range(2))
    eng.time_batch(codec[0], 1, db, 1); td = min(eng.time_batch(codec[0], 1, db, 1) for _ in range(2))
    print(f"{kind:10s} ratio {total / int(clen.sum().item()):6.3f}  compress {total / (1 << 30) / (tc * 1e-3):7.1f} GiB/s  decompress {total / (1 << 30) / (td * 1e-3):7.1f} GiB/s", flush=True)
    del src, slots, dst
