#!/usr/bin/env python3
"""Experiment: per-stage cycle breakdown of lz4_compress_kernel.  Needs a library built with
AOCLGPU_CFLAGS="-DAOCLGPU_EXPERIMENTS" (the release library ignores the switch and exports no aoclgpu_debug_* symbol)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aocl_compression_amd as a
eng = a.engine(0)
data = a.corpus.make(sys.argv[1], 64 << 20, seed=7) if len(sys.argv) > 1 else a.corpus.silesia_mix(64 << 20)     # [kind]: one corpus kind instead of the mix
src = torch.from_numpy(data).to(eng.device).repeat(16)
chunk = 262144
slots, out_len, slot = eng.compress_chunks(a.LZ4, src, chunk, flags=8)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 16)()
eng.lib.aoclgpu_debug_lz4_prof(buf)
names = ["ring + probe windows + backfill", "hash + table lookup", "cand loads + slot-reuse", "cand<4 fix-up + dups + fw/bk", "walk + anchors + commit", "records + state",
         "one-at-a-time path", "general batches + flush"]
tot = sum(buf[:8]); c = list(buf[8:16]); nseq = max(c[1], 1)
for n, v in zip(names, buf[:8]):
    print(f"{n:46s} {v/tot*100:6.2f}%  {v/nseq:8.1f} cyc/seq  {v/max(c[0],1):8.1f} cyc/batch")
if hasattr(eng.lib, "aoclgpu_debug_lz4_prof2"):           # -DLZ4_PROF_SPLIT
    b2 = (C.c_ulonglong * 8)(); eng.lib.aoclgpu_debug_lz4_prof2(b2)
    if sum(b2):
        for n, v in zip(["  (split) walk", "  (split) anchors + cover", "  (split) commit = 'walk + anchors + commit' above"], [b2[0], b2[1], buf[4]]):
            print(f"{n:46s} {v/(tot+sum(b2))*100:6.2f}%  {v/nseq:8.1f} cyc/seq  {v/max(c[0],1):8.1f} cyc/batch")
print("total cyc/seq", tot / nseq, " cyc/batch", tot / max(c[0], 1))
print("batches", c[0], "(general", c[2], ", consecutive batches of a continuing search", c[3], ") sequences", c[1], "seq/batch", c[1] / max(c[0], 1),
      "avg lim", c[6] / max(c[0], 1), "bytes/batch", src.numel() / max(c[0], 1))
print("one-at-a-time sequences", c[4], " record flushes", c[5], " sequences per flush", c[1] / max(c[5], 1))
