import sys, ctypes as C, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import aocl_compression_amd as a, oracle_lib as ol
eng = a.engine(0); ref = ol.Ref()
ref.L.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
rng = np.random.default_rng(3)
pools = {k: a.corpus.make(k, 1 << 20, seed=6).tobytes() for k in ("text", "jsonlog", "binary", "database", "image16")}
for it in range(60):
    k = list(pools)[it % len(pools)]
    size = int(rng.choice([0, 1, 100, 5000, 70000, 131072, 300000, 600000]))
    off = int(rng.integers(0, len(pools[k]) - size))
    raw = pools[k][off:off + size]
    level = int(rng.choice([1, 1, 3, 5, 9, 19, -3]))
    out = C.create_string_buffer(len(raw) + len(raw) // 200 + 1024)
    r = ref.L.ZSTD_compress(out, len(out), raw, len(raw), level)
    dst = C.create_string_buffer(max(size, 1))
    rr = eng.lib.aoclgpu_decompress_host(a.ZSTD, out.raw[:r], r, dst, max(size, 1))
    if rr != size or dst.raw[:size] != raw:
        d = np.frombuffer(dst.raw[:size], np.uint8); w = np.frombuffer(raw, np.uint8)
        ne = np.nonzero(d != w)[0]
        # block structure
        s = out.raw[:r]; fhd = s[4]; pos = 5 + (0 if fhd & 0x20 else 1) + [0,1,2,4][fhd & 3] + ([1,2,4,8][fhd >> 6] if (fhd >> 6) or (fhd & 0x20) else 0)
        blocks = []
        while pos + 3 <= len(s):
            h = s[pos] | (s[pos+1] << 8) | (s[pos+2] << 16); last = h & 1; bt = (h >> 1) & 3; bs = h >> 3
            lit = s[pos+3] if bt == 2 else None
            blocks.append((bt, bs, None if lit is None else (lit & 3, (lit >> 2) & 3)))
            pos += 3 + (1 if bt == 1 else bs)
            if last: break
        print(it, k, size, off, level, "rc", rr, "first diff", ne[:3] if len(ne) else None, blocks)
print("done")
