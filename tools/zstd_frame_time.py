"""Throughput of whole-input Zstd frames (aoclgpu_zstd_compress_frames, one wavefront per frame):
   python tools/zstd_frame_time.py            # one frame alone at several sizes, then batches of frames side by side
Prints one JSON line per case.  Data already in HBM, level-1 parameters of the size (AOCL table)."""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import aocl_compression_amd as a

eng = a.engine(0)
base = torch.from_numpy(a.corpus.silesia_mix(64 << 20)).to(eng.device)


def params(n):
    src_log = 6 if n < 64 else int(n - 1).bit_length()
    if n <= 16384: w, h, m = 14, 15, 5
    elif n <= 131072: w, h, m = 17, 13, 6
    elif n <= 262144: w, h, m = 18, 14, 6
    else: w, h, m = 23, 14, 7
    w = min(w, src_log)
    return w, min(h, w + 1), m


def run(frame_bytes, nframes, reps=3):
    total = frame_bytes * nframes
    src = base.repeat((total + base.numel() - 1) // base.numel())[:total].contiguous()
    off = np.arange(nframes + 1, dtype=np.uint64) * frame_bytes
    w, h, m = params(frame_bytes)
    best = 1e9
    for it in range(reps + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        dst, doff, lens = eng.zstd_compress_frames(src, off, w, h, m)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if it: best = min(best, dt)
    assert all(x > 0 for x in lens)
    print(json.dumps({"frame_bytes": frame_bytes, "frames": nframes, "ms": round(best * 1e3, 2), "MiB_per_s": round(total / best / 2**20, 1),
                      "ratio": round(total / sum(lens), 3), "params": [w, h, m]}), flush=True)


for fb in (128 << 10, 1 << 20, 16 << 20):
    run(fb, 1)
for fb, k in ((1 << 20, 64), (1 << 20, 1024), (4 << 20, 1024), (16 << 20, 256)):
    run(fb, k)
