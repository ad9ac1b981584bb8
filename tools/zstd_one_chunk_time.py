import sys, time, numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import aocl_compression_amd as a
eng = a.engine(0)
base = torch.from_numpy(a.corpus.silesia_mix(64 << 20)).to(eng.device)
for n in (131072,):
    src = base[:n].contiguous()
    for name, fn in (("chunk kernel", lambda: eng.compress_chunks(a.ZSTD, src, n)),
                     ("frame kernel", lambda: eng.zstd_compress_frames(src, [0, n], 17, 13, 6))):
        best = 1e9
        for it in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            if it: best = min(best, dt)
        print(name, n, round(best * 1e3, 2), "ms")
# 64 chunks / frames side by side (still far below the chip's slots): per-wave speed without contention
n = 131072; k = 64
src = base[:n * k].contiguous()
for name, fn in (("chunk kernel x64", lambda: eng.compress_chunks(a.ZSTD, src, n)),
                 ("frame kernel x64", lambda: eng.zstd_compress_frames(src, np.arange(k + 1) * n, 17, 13, 6))):
    best = 1e9
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if it: best = min(best, dt)
    print(name, round(best * 1e3, 2), "ms")
