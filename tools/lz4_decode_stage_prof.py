#!/usr/bin/env python3
"""Experiment: per-stage cycle breakdown of lz4_decompress_kernel.  Needs a library built with
AOCLGPU_CFLAGS="-DAOCLGPU_EXPERIMENTS -DLZ4D_PROF" (python aocl-compression_amd/build.py --variant dprof -DAOCLGPU_EXPERIMENTS -DLZ4D_PROF)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aocl_compression_amd as a
eng = a.engine(0)
data = a.corpus.make(sys.argv[1], 64 << 20, seed=7) if len(sys.argv) > 1 else a.corpus.silesia_mix(64 << 20)     # [kind]: one corpus kind instead of the mix
src = torch.from_numpy(data).to(eng.device).repeat(16)
chunk = 262144
slots, out_len, slot = eng.compress_chunks(a.LZ4, src, chunk)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 16)()
eng.lib.aoclgpu_debug_lz4_prof(buf)          # clear
dst, dl = eng.decompress_chunks(a.LZ4, slots, out_len, slot, src.numel(), chunk)
torch.cuda.synchronize()
eng.lib.aoclgpu_debug_lz4_prof(buf)
names = ["input staging", "scan: parse", "scan: walk + records", "group: setup + requests", "group: wait + block writes", "group: matches inside the block",
         "group: block out + history", "sequential path"]
tot = sum(buf[:8]); c = list(buf[8:16])
for n, v in zip(names, buf[:8]):
    print(f"{n:34s} {v/tot*100:6.2f}%  {v/max(c[1],1):8.1f} cyc/seq")
print("total cyc/seq", tot / max(c[1], 1))
print("scan iterations", c[0], "sequences recorded", c[1], "per iteration", c[1] / max(c[0], 1), " groups", c[2], "sequences per group", c[3] / max(c[2], 1),
      "in-block matches per group", c[4] / max(c[2], 1), " sequential steps", c[5])
