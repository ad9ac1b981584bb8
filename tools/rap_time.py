import sys, time, ctypes as C, torch
sys.path.insert(0, ".")
import aocl_compression_amd as a
eng = a.engine(0)
base = a.corpus.silesia_mix(64 << 20)
src = torch.from_numpy(base).to(eng.device).repeat(16)
N = src.numel()
for name, codec, chunk in (("lz4", a.LZ4, 262144), ("snappy", a.SNAPPY, 65536), ("zstd", a.ZSTD, 131072)):
    t = torch
    slot = (a.compress_bound(codec, chunk) + 15) & ~15
    nchunks = N // chunk
    slots = t.empty(nchunks * slot, dtype=t.uint8, device=eng.device)
    out_len = t.zeros(nchunks, dtype=t.int32, device=eng.device)
    aux = t.zeros(nchunks, dtype=t.int32, device=eng.device)
    b = eng.uniform_compress_batch(src, chunk, slots, out_len, slot)
    b.aux = aux.data_ptr()
    cap = int(eng.lib.aoclgpu_rap_frame_bytes(codec, nchunks)) + nchunks * slot + 16
    out = t.empty(cap, dtype=t.uint8, device=eng.device)
    n = C.c_uint64(0)
    tc = tp = 0
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng._check(eng._fn(codec, "compress")(C.byref(b), eng._stream()), "c")
        torch.cuda.synchronize(); t1 = time.perf_counter()
        eng._check(eng.lib.aoclgpu_rap_pack(codec, C.byref(b), out.data_ptr(), cap, C.byref(n), eng._stream()), "p")
        torch.cuda.synchronize(); t2 = time.perf_counter()
        if it: tc += t1 - t0; tp += t2 - t1
    print(name, "compress ms", round(tc / 3 * 1e3, 2), "pack ms", round(tp / 3 * 1e3, 2), "GiB/s c", round(1 / (tc / 3), 1), "c+p", round(1 / ((tc + tp) / 3), 1))
