#!/usr/bin/env python3
"""Experiment: per-stage cycle breakdown of the Zstd match finder and decoder.  Needs a library built with
AOCLGPU_CFLAGS="-DAOCLGPU_EXPERIMENTS -DZSTD_PROF"; ZEXP=16 skips the entropy stage (raw blocks), ZEXP=32 disables chain batches."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aocl_compression_amd as a
eng = a.engine(0)
data = a.corpus.make(sys.argv[1], 64 << 20, seed=7) if len(sys.argv) > 1 else a.corpus.silesia_mix(64 << 20)     # [kind]: one corpus kind instead of the mix
src = torch.from_numpy(data).to(eng.device).repeat(8)
ZEXP = int(os.environ.get("ZEXP", "0"))
slots, out_len, slot = eng.compress_chunks(a.ZSTD, src, 131072, flags=ZEXP)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 16)()
eng.lib.aoclgpu_debug_zstd_prof(buf)
names = ["probe loads (ld64 at pos)", "hash + table + cand/rep loads", "slot-reuse rounds", "decision + commit (+ no-hit batches)", "ip1 write + catch-up + extension", "store seq + post writes + rep2 / loop top"]
names += ["chain batch: equality masks (waits for the candidates)", "chain batch: the scalar walk alone"]
tot = sum(buf[:8]); nseq = buf[9]
for n, v in zip(names, buf[:8]):
    print(f"{n:44s} {v/tot*100:6.2f}%  {v/nseq:8.1f} cyc/seq")
print("total cyc/seq", tot / nseq, "batches", buf[8], "seqs", nseq, "batches/seq", buf[8] / nseq)
print("chain outcomes: restart %d continue %d event %d | one-match batches %d | immediate repcodes %d | map refreshes %d" % tuple(buf[10:16]))
dst, dl = eng.decompress_chunks(a.ZSTD, slots, out_len, slot, src.numel(), 131072)
torch.cuda.synchronize()
eng.lib.aoclgpu_debug_zstd_dprof(buf)
dn = ["literals section (Huffman table + 4 streams)", "sequence tables + bitstream init", "FSE sequence decode (lane 0)", "prefix sums + literal copies", "match copies (in order)"]
tot = sum(buf[:8])
dn += ["lit: before the streams (table)", "lit: speculative count rounds", "lit: writing run"]
print("---- decoder (entropy stage skipped on the compress side -> raw blocks!)" if ZEXP == 16 else "---- decoder")
for n, v in zip(dn, buf[:8]):
    print(f"{n:48s} {v/tot*100:6.2f}%")
c = list(buf[8:16])
if c[0]:
    print("tiles %d  sequences/tile %.1f  match levels/tile %.2f  blocked-loop trips/tile %.1f  whole-wave literal runs/tile %.2f  whole-wave matches/tile %.2f  one-per-lane matches/tile %.1f"
          % (c[0], c[5] / c[0], c[1] / c[0], c[2] / c[0], c[3] / c[0], c[4] / c[0], c[6] / c[0]))
