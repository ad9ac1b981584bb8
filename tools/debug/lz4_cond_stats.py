import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import aocl_compression_amd as m
eng = m.engine(0)
for kind in ("text", "binary", "database", "image16", "mix"):
    base = torch.from_numpy(m.corpus.silesia_mix(64 << 20, seed=7) if kind == "mix" else m.corpus.make(kind, 64 << 20, seed=7)).to(eng.device)
    chunk = 262144; n = base.numel() // chunk
    slot = (m.compress_bound(m.LZ4, chunk) + 15) & ~15
    slots = torch.empty(n * slot, dtype=torch.uint8, device=eng.device); clen = torch.zeros(n, dtype=torch.int32, device=eng.device)
    cb = eng.uniform_compress_batch(base[:n * chunk].contiguous(), chunk, slots, clen, slot)
    buf = (C.c_ulonglong * 8)()
    eng.lib.aoclgpu_debug_lz4_cond(buf)
    eng._check(eng._fn(m.LZ4, "compress")(C.byref(cb), eng._stream()), "c"); torch.cuda.synchronize()
    eng.lib.aoclgpu_debug_lz4_cond(buf)
    print(kind, "could-match dups", buf[0], "conditional", buf[1], "stops with the slot written", buf[2], "not written", buf[3], "no hit", buf[4], "| batches", buf[7], "cut by a third lane", buf[5], "mean lim %.1f" % (buf[6] / max(buf[7], 1)))
