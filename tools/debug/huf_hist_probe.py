"""bytes with a given histogram (tools/debug/hist_case1.txt: the literals of a block on which the frame kernel's Huffman table
differed from the reference's), shuffled: chunk kernel against the oracle"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import aocl_compression_amd as a
import oracle_lib as ol
eng = a.engine(0); orc = ol.Oracle()
h = np.array([int(x) for x in open(os.path.join(ROOT, "tools", "debug", "hist_case1.txt"))])
base = np.repeat(np.arange(256, dtype=np.uint8), h)
bad = 0
for seed in range(20):
    rng = np.random.default_rng(seed)
    data = rng.permutation(base).tobytes()
    want = orc.compress(ol.ZSTD, data)
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(eng.device)
    slots, lens, slot = eng.compress_chunks(a.ZSTD, t, len(data))
    got = slots[:int(lens[0])].cpu().numpy().tobytes()
    if got != want:
        bad += 1; print("MISMATCH seed", seed, len(got), len(want))
print("bad", bad, "of 20")

# code lengths of the last case: GPU table against the oracle's
import ctypes as C
orc.L.orc_dbg_huf_lengths.restype = C.c_uint
orc.L.orc_dbg_huf_lengths.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]


def lengths(frame):
    body = frame[4 + 1 + (4 if len(data) >= 65536 + 256 else 2) + 3:]   # magic, FHD, FCS, block header
    assert (body[0] & 3) == 2
    lh = 3 + ((body[0] >> 2) & 3 >= 2) + ((body[0] >> 2) & 3 == 3)
    out = C.create_string_buffer(256)
    tl = orc.L.orc_dbg_huf_lengths(body[lh:lh + 200], 200, out)
    return tl, list(out.raw)


tg, lg = lengths(got); tw, lw = lengths(want)
print("table log gpu", tg, "oracle", tw)
diff = [(s, int(h[s]), lg[s], lw[s]) for s in range(256) if lg[s] != lw[s]]
print("symbols whose length differs (symbol, count, gpu, oracle):", diff)
from collections import Counter
print("gpu   length histogram", sorted(Counter(lg).items())); print("oracle length histogram", sorted(Counter(lw).items()))
print("cost gpu", sum(int(h[s]) * lg[s] for s in range(256)), "oracle", sum(int(h[s]) * lw[s] for s in range(256)))
