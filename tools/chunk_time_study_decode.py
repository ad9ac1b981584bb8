#!/usr/bin/env python3
"""Experiment (library built with AOCLGPU_CFLAGS=-DAOCLGPU_EXPERIMENTS): per-chunk durations of lz4_decompress_kernel /
snappy-like decoders on the bench corpus, how well the compressed size predicts them, and what a launch order would do to the
tail (list-scheduling simulation on the measured durations: `slots` resident chunks, started in index order vs longest-first
vs ordered by compressed size)."""
import ctypes as C, heapq, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aocl_compression_amd as a
eng = a.engine(0)
base = torch.from_numpy(a.corpus.silesia_mix(211938580, seed=0xA0C1)).to(eng.device)
chunk = 262144
total = (8 << 30)
src = base.repeat(-(-total // base.numel()))[:total].contiguous()
n = total // chunk
slot = (a.compress_bound(a.LZ4, chunk) + 15) & ~15
slots = torch.empty(n * slot, dtype=torch.uint8, device=eng.device)
clen = torch.zeros(n, dtype=torch.int32, device=eng.device)
cb = eng.uniform_compress_batch(src, chunk, slots, clen, slot)
eng._check(eng._fn(a.LZ4, "compress")(C.byref(cb), eng._stream()), "c")
dst = torch.empty(total, dtype=torch.uint8, device=eng.device)
dlen = torch.zeros(n, dtype=torch.int32, device=eng.device)
aux = torch.zeros(n, dtype=torch.int32, device=eng.device)
db = eng.uniform_decompress_batch(slots, clen, slot, dst, chunk, dlen)
db.aux = aux.data_ptr(); db.reserved = 2
for _ in range(2):
    eng._check(eng._fn(a.LZ4, "decompress")(C.byref(db), eng._stream()), "d")
torch.cuda.synchronize()
t = aux.cpu().numpy().astype(np.float64)
cl = clen.cpu().numpy().astype(np.float64)
print("chunks", n, "mean", t.mean(), "max", t.max(), "p99", np.percentile(t, 99), "min", t.min(), "max/mean", t.max() / t.mean())
print("corr(time, compressed size)", np.corrcoef(t, cl)[0, 1])
r = cl / chunk
for lo in np.arange(0, 1.01, 0.1):
    m = (r >= lo) & (r < lo + 0.1)
    if m.any(): print(f"ratio {lo:.1f}-{lo+0.1:.1f}: {m.sum():6d} chunks, mean time {t[m].mean():8.1f}, max {t[m].max():8.1f}")
def simulate(order, nslots):
    h = [0.0] * nslots
    heapq.heapify(h)
    end = 0.0
    for i in order:
        s_ = heapq.heappop(h); e = s_ + t[i]; end = max(end, e); heapq.heappush(h, e)
    return end
for nslots in (6144, 8192):
    ideal = t.sum() / nslots
    res = {"slots": nslots, "ideal": ideal, "in_order": simulate(range(n), nslots) / ideal, "longest_first": simulate(np.argsort(-t), nslots) / ideal,
           "by_compressed_size_desc": simulate(np.argsort(-cl, kind="stable"), nslots) / ideal, "by_compressed_size_asc": simulate(np.argsort(cl, kind="stable"), nslots) / ideal}
    bucket = np.minimum((r * 20).astype(int), 19)
    bm = np.array([t[bucket == k].mean() if (bucket == k).any() else 0 for k in range(20)])
    res["by_ratio_bucket_mean_time"] = simulate(np.argsort(-bm[bucket], kind="stable"), nslots) / ideal
    print(json.dumps(res))
