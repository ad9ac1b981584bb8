import sys, os, collections, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import aocl_compression_amd as a
from aocl_compression_amd import api
import oracle_lib as ol
eng = a.engine(0)
L = api.load_library()
L.ZSTD_decompress.restype = C.c_size_t
L.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
ref = ol.Ref()
rng = np.random.default_rng(62)
cnt = collections.Counter(); ex = {}
def muts(comp, count, short):
    yield comp, "intact"
    for cut in (1, 2, 3, 5, 17, len(comp) // 2):
        if cut < len(comp): yield comp[:-cut], f"trunc{cut}"
    for it in range(count):
        c = bytearray(comp[:short] if it % 2 else comp)
        for _ in range(1 + it % 3): c[int(rng.integers(0, len(c)))] = int(rng.integers(0, 256))
        yield bytes(c), f"mut{it}"
for kind, size in (("binary", 131072), ("text", 70000), ("jsonlog", 100000), ("random", 3000), ("database", 5000)):
    raw = a.corpus.make(kind, size, seed=12).tobytes()
    comp = ref.compress(ol.ZSTD, raw)
    for c, what in muts(comp, 100, 600):
        for cap in (len(raw), len(raw) - 1):
            w = C.create_string_buffer(max(cap, 1)); g = C.create_string_buffer(max(cap, 1))
            want = ref.L.ZSTD_decompress(w, cap, c, len(c)); got = L.ZSTD_decompress(g, cap, c, len(c))
            ws = -(want - (1 << 64)) if want > (1 << 63) else "ok"; gs = -(got - (1 << 64)) if got > (1 << 63) else "ok"
            cnt[(ws, gs)] += 1
            ex.setdefault((ws, gs), (kind, what, cap))
for k, v in sorted(cnt.items(), key=lambda x: -x[1]): print(k, v, ex[k])
# details of the accept / reject differences
import base64
rng = np.random.default_rng(62)
shown = 0
for kind, size in (("binary", 131072), ("text", 70000), ("jsonlog", 100000), ("random", 3000), ("database", 5000)):
    raw = a.corpus.make(kind, size, seed=12).tobytes()
    comp = ref.compress(ol.ZSTD, raw)
    for c, what in muts(comp, 100, 600):
        cap = len(raw)
        w = C.create_string_buffer(max(cap, 1)); g = C.create_string_buffer(max(cap, 1))
        want = ref.L.ZSTD_decompress(w, cap, c, len(c)); got = L.ZSTD_decompress(g, cap, c, len(c))
        if want < (1 << 63) and got > (1 << 63) and shown < 12:
            diff = [i for i in range(min(len(c), len(comp))) if c[i] != comp[i]]
            print(kind, what, "len", len(c), "of", len(comp), "ref returns", want, "diff at", diff, "hdr", c[:12].hex(), "eng.decompress_host:", eng.decompress_host(a.ZSTD, c, cap) is not None)
            shown += 1
