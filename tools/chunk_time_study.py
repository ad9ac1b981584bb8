#!/usr/bin/env python3
"""Experiment (library built with AOCLGPU_CFLAGS=-DAOCLGPU_EXPERIMENTS): per-chunk durations of lz4_compress_kernel on the
bench corpus, how well the compressed size predicts them, and what launch order would do to the tail (list-scheduling
simulation on the measured durations: 4096 slots, chunks started in index order vs longest-first vs a proxy order)."""
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
out_len = torch.zeros(n, dtype=torch.int32, device=eng.device)
aux = torch.zeros(n, dtype=torch.int32, device=eng.device)
b = eng.uniform_compress_batch(src, chunk, slots, out_len, slot)
b.aux = aux.data_ptr(); b.reserved = 2
for _ in range(2):
    eng._check(eng._fn(a.LZ4, "compress")(C.byref(b), eng._stream()), "c")
torch.cuda.synchronize()
t = aux.cpu().numpy().astype(np.float64)        # ticks / 1024 (100 MHz s_memtime -> ~10 us units)
cl = out_len.cpu().numpy().astype(np.float64)
print("chunks", n, "mean", t.mean(), "max", t.max(), "p99", np.percentile(t, 99), "min", t.min(), "max/mean", t.max() / t.mean())
print("corr(time, compressed size)", np.corrcoef(t, cl)[0, 1])
# time by ratio bucket
r = cl / chunk
for lo in np.arange(0, 1.01, 0.1):
    m = (r >= lo) & (r < lo + 0.1)
    if m.any(): print(f"ratio {lo:.1f}-{lo+0.1:.1f}: {m.sum():6d} chunks, mean time {t[m].mean():8.1f}, max {t[m].max():8.1f}")
def simulate(order, slots=4096, t=None):
    t = globals()["t"] if t is None else t
    h = [0.0] * slots
    heapq.heapify(h)
    end = 0.0
    for i in order:
        s_ = heapq.heappop(h); e = s_ + t[i]; end = max(end, e); heapq.heappush(h, e)
    return end
ideal = t.sum() / 4096
inorder = simulate(range(n))
lpt = simulate(np.argsort(-t))
# proxy: mean time of the chunk's ratio bucket (what a table learnt once per codec could give)
bucket = np.minimum((r * 20).astype(int), 19)
bm = np.array([t[bucket == k].mean() if (bucket == k).any() else 0 for k in range(20)])
proxy = simulate(np.argsort(-bm[bucket], kind="stable"))
print(json.dumps({"ideal": ideal, "in_order": inorder, "longest_first": lpt, "by_ratio_bucket": proxy,
                  "in_order_over_ideal": inorder / ideal, "lpt_over_ideal": lpt / ideal, "proxy_over_ideal": proxy / ideal}))

# ---- decoder: durations vs the (known in advance) compressed size
dst = torch.empty(total, dtype=torch.uint8, device=eng.device)
dlen = torch.zeros(n, dtype=torch.int32, device=eng.device)
daux = torch.zeros(n, dtype=torch.int32, device=eng.device)
b.reserved = 0
eng._check(eng._fn(a.LZ4, "compress")(C.byref(b), eng._stream()), "c")
db = eng.uniform_decompress_batch(slots, out_len, slot, dst, chunk, dlen)
db.aux = daux.data_ptr(); db.reserved = 2
for _ in range(2):
    eng._check(eng._fn(a.LZ4, "decompress")(C.byref(db), eng._stream()), "d")
torch.cuda.synchronize()
td = daux.cpu().numpy().astype(np.float64)
print("decode: mean", td.mean(), "max", td.max(), "min", td.min(), "max/mean", td.max() / td.mean(), "corr(time, csize)", np.corrcoef(td, cl)[0, 1])
for lo in np.arange(0, 1.01, 0.1):
    m = (r >= lo) & (r < lo + 0.1)
    if m.any(): print(f"ratio {lo:.1f}-{lo+0.1:.1f}: {m.sum():6d} chunks, mean decode time {td[m].mean():8.1f}, max {td[m].max():8.1f}")
S = 8192
bmd = np.array([td[bucket == k].mean() if (bucket == k).any() else 0 for k in range(20)])
print(json.dumps({"decode_ideal": td.sum() / S, "in_order": simulate(range(n), S, td), "longest_first": simulate(np.argsort(-td), S, td),
                  "by_ratio_bucket": simulate(np.argsort(-bmd[bucket], kind="stable"), S, td), "by_csize_desc": simulate(np.argsort(-cl, kind="stable"), S, td),
                  "by_csize_asc": simulate(np.argsort(cl, kind="stable"), S, td)}))

np.savez_compressed(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "chunk_times.npz"), t=t, td=td, cl=cl)
