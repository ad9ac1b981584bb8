"""i.i.d. bytes from skewed 256-symbol distributions (no matches, Huffman-compressible): stresses the Huffman sort / tree
builder with large buckets of similar counts.  Chunk kernel (level 1) against the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import aocl_compression_amd as a
import oracle_lib as ol
eng = a.engine(0); orc = ol.Oracle()
bad = tot = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 200):
    rng = np.random.default_rng(seed)
    kind = seed % 4
    if kind == 0: w = rng.random(256) ** rng.uniform(0.3, 4)
    elif kind == 1: w = np.exp(-np.arange(256) / rng.uniform(5, 200)); rng.shuffle(w)
    elif kind == 2: w = 1 + rng.random(256) * rng.uniform(0.01, 1.0)
    else: w = np.r_[np.full(int(rng.integers(1, 8)), 40.0), 1 + rng.random(256)][:256]
    w = w / w.sum()
    n = int(rng.choice([131072, 131072, 65536, 100000, 20000]))
    data = rng.choice(256, size=n, p=w).astype(np.uint8).tobytes()
    want = orc.compress(ol.ZSTD, data)
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(eng.device)
    slots, lens, slot = eng.compress_chunks(a.ZSTD, t, n)
    got = slots[:int(lens[0])].cpu().numpy().tobytes()
    tot += 1
    if got != want:
        bad += 1
        print("MISMATCH seed", seed, "kind", kind, "n", n, len(got), len(want))
print("total", tot, "bad", bad)
