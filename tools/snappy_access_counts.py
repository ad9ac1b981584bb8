#!/usr/bin/env python3
"""Experiment (library built with AOCLGPU_CFLAGS="-DAOCLGPU_EXPERIMENTS -DSN_COUNT"): memory accesses of the Snappy parser by
origin on the bench stream (config 3: 64 KiB chunks of the JSON / log corpus), LDS-table waves and HBM-table waves of the mixed
kernel apart, next to what rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE report for the same launch (profiles/traffic.json)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aocl_compression_amd as a
eng = a.engine(0)
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
base = torch.from_numpy(a.corpus.make("jsonlog", 211938580, seed=0xA0C3)).to(eng.device)
chunk = 65536
total = int(gib * (1 << 30)) // chunk * chunk
src = base.repeat(-(-total // base.numel()))[:total].contiguous()
n = total // chunk
slot = (a.compress_bound(a.SNAPPY, chunk) + 15) & ~15
slots = torch.empty(n * slot, dtype=torch.uint8, device=eng.device)
clen = torch.zeros(n, dtype=torch.int32, device=eng.device)
cb = eng.uniform_compress_batch(src, chunk, slots, clen, slot)
buf = (C.c_ulonglong * 16)()
eng.lib.aoclgpu_debug_snappy_counts(buf)
eng._check(eng._fn(a.SNAPPY, "compress")(C.byref(cb), eng._stream()), "c")
torch.cuda.synchronize()
eng.lib.aoclgpu_debug_snappy_counts(buf)
names = ["batches", "probe-window loads (lanes, 2 x 16 B each)", "table slot reads (lanes)", "candidate-window loads (lanes, 2 x 16 B each)",
         "table slot writes (lanes)", "table bytes filled", "extension-loop rounds (64 x 2 x 4 B each)"]
out = {"gib": gib, "chunks": n, "compressed_bytes": int(clen.sum().item())}
for kind, o in (("lds_table_waves", 0), ("hbm_table_waves", 8)):
    out[kind] = {nm: int(buf[o + k]) for k, nm in enumerate(names)}
h = out["hbm_table_waves"]
# an upper bound on what the HBM-table waves ask of the memory system if every access missed every cache (64-byte sectors)
out["hbm_table_waves_sector_bound_GB"] = {
    "slot reads x 64 B": h[names[2]] * 64 / 1e9, "candidate windows x 64 B (one sector, sometimes two)": h[names[3]] * 64 / 1e9,
    "slot writes x 64 B (write-back of a partly written sector)": h[names[4]] * 64 / 1e9, "table fill": h[names[5]] / 1e9,
    "probe windows x 32 B (consecutive lanes share sectors: mostly L1 hits)": h[names[1]] * 32 / 1e9}
print(json.dumps(out, indent=1))
