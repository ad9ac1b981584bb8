#!/usr/bin/env python3
"""Measurement (GPU box): Zstd level 2 compress throughput on the bench corpus (2 GiB, 128 KiB chunks), round trip checked."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aocl_compression_amd as a
eng = a.engine(0)
base = torch.from_numpy(a.corpus.silesia_mix(128 << 20)).to(eng.device)
src = base.repeat(16)
for name, flags in (("level 1", 0), ("level 2", a.api.FLAG_ZSTD_LEVEL2)):
    eng.compress_chunks(a.ZSTD, src[:64 << 20], 131072, flags=flags)
    torch.cuda.synchronize(); t0 = time.time()
    slots, out_len, slot = eng.compress_chunks(a.ZSTD, src, 131072, flags=flags)
    torch.cuda.synchronize(); dt = time.time() - t0
    dst, dl = eng.decompress_chunks(a.ZSTD, slots, out_len, slot, src.numel(), 131072)
    assert torch.equal(dst, src)
    print(f"zstd {name}: {src.numel() / dt / 2**30:.2f} GiB/s compress (wall, incl. allocation), ratio {src.numel() / int(out_len.sum()):.3f}")
