#!/usr/bin/env python3
"""Experiment: per-stage cycle breakdown of the double-fast match finder (level 3).  Needs a library built with
-DAOCLGPU_EXPERIMENTS -DZSTD_PROF (python aocl-compression_amd/build.py --variant prof ...), loaded through AOCLGPU_LIB."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aocl_compression_amd as a
eng = a.engine(0)
data = a.corpus.silesia_mix(64 << 20)
src = torch.from_numpy(data).to(eng.device).repeat(4)
buf = (C.c_ulonglong * 16)()
eng.lib.aoclgpu_debug_zstd_prof(buf)
slots, out_len, slot = eng.compress_chunks(a.ZSTD, src, 131072, flags=a.api.zstd_level(3))
torch.cuda.synchronize()
eng.lib.aoclgpu_debug_zstd_prof(buf)
names = ["probe windows (ld64 at pos)", "two table reads", "candidates + repcode bytes", "slot-reuse rounds + first event", "commit + event: extension round", "sequence + insertions + rep2 loop"]
tot = sum(buf[:6]); nseq = buf[9]
for n, v in zip(names, buf[:6]):
    print(f"{n:40s} {v/tot*100:6.2f}%  {v/nseq:8.1f} cyc/seq")
print("total cyc/seq", tot / nseq, "batches", buf[8], "first", buf[10], "further", buf[11], "seqs", nseq, "batches/seq", buf[8] / nseq)
