#!/usr/bin/env python3
"""Kernel A/B harness (development tool): times the compress and decompress launches of one codec on the bench workload for
several builds of the library on the SAME box, and prints a checksum of the compressed stream so that a faster kernel that
changes one byte is seen at once.

    python tools/kbench.py --codec lz4 [--gib 8] [--iters 3] [lib.so ...]      (no library = the product library)

Every library runs in its own child process (AOCLGPU_LIB, see aocl-compression_amd/build.py --variant)."""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CODECS = {"lz4": (0, 262144), "snappy": (1, 65536), "zstd": (2, 131072)}


def child(a):
    import numpy as np, torch
    import aocl_compression_amd as m
    eng = m.engine(0)
    codec, chunk = CODECS[a.codec]
    chunk = a.chunk or chunk
    total = int(a.gib * (1 << 30)) // chunk * chunk
    nbase = 211_938_580
    base_np = m.corpus.make("jsonlog", nbase, seed=0xA0C3) if a.codec == "snappy" else m.corpus.silesia_mix(nbase, seed=0xA0C1)
    base = torch.from_numpy(base_np).to(eng.device)
    src = base.repeat(-(-total // base.numel()))[:total].contiguous()
    n = total // chunk
    slot = (m.compress_bound(codec, chunk) + 15) & ~15
    slots = torch.empty(n * slot, dtype=torch.uint8, device=eng.device)
    clen = torch.zeros(n, dtype=torch.int32, device=eng.device)
    dst = torch.empty(total, dtype=torch.uint8, device=eng.device)
    dlen = torch.zeros(n, dtype=torch.int32, device=eng.device)
    cb = eng.uniform_compress_batch(src, chunk, slots, clen, slot)
    cb.reserved = a.flags
    db = eng.uniform_decompress_batch(slots, clen, slot, dst, chunk, dlen)
    eng.time_batch(codec, 0, cb, 1)
    tc = min(eng.time_batch(codec, 0, cb, 1) for _ in range(a.iters))
    eng.time_batch(codec, 1, db, 1)
    td = min(eng.time_batch(codec, 1, db, 1) for _ in range(a.iters))
    ok = bool(torch.equal(dst, src)) and int((clen <= 0).sum().item()) == 0
    # checksum of the compressed stream: per-chunk weighted byte sums over the used part of every slot
    csum = 0
    step = 4096
    w = (torch.arange(slot, device=eng.device, dtype=torch.int64) % 251 + 1)
    for i in range(0, n, step):
        k = min(step, n - i)
        blk = slots[i * slot:(i + k) * slot].view(k, slot).to(torch.int64)
        mask = torch.arange(slot, device=eng.device).unsqueeze(0) < clen[i:i + k].unsqueeze(1)
        csum = (csum * 1000003 + int(((blk * w) * mask).sum().item())) % (1 << 61)
    print(json.dumps({"lib": os.path.basename(os.environ.get("AOCLGPU_LIB", "product")), "codec": a.codec, "gib": a.gib,
                      "compress_ms": round(tc, 3), "decompress_ms": round(td, 3),
                      "compress_GiBps": round(total / (1 << 30) / (tc * 1e-3), 2), "decompress_GiBps": round(total / (1 << 30) / (td * 1e-3), 2),
                      "bytes": int(clen.sum().item()), "checksum": csum, "round_trip": ok}), flush=True)


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("--codec", default="lz4", choices=list(CODECS))
    p.add_argument("--gib", type=float, default=8.0)
    p.add_argument("--chunk", type=int, default=0)
    p.add_argument("--iters", type=int, default=3)
    p.add_argument("--flags", type=int, default=0)
    p.add_argument("--child", action="store_true")
    p.add_argument("libs", nargs="*")
    a = p.parse_args()
    if a.child:
        child(a)
    else:
        for lib in (a.libs or [""]):
            env = dict(os.environ)
            if lib:
                env["AOCLGPU_LIB"] = os.path.abspath(lib)
            else:
                env.pop("AOCLGPU_LIB", None)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--codec", a.codec, "--gib", str(a.gib), "--chunk", str(a.chunk),
                            "--iters", str(a.iters), "--flags", str(a.flags)], env=env, check=False)
