#!/usr/bin/env python3
"""Host-side probe for bench.py's cpu_baseline: what CPU budget does this container really have (cgroup quota, cpuset)
and how does the reference LZ4 path scale with OpenMP threads here?  TEST INFRASTRUCTURE (loads oracle/)."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpuset.cpus.effective", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, "=", open(f).read().strip())
    except OSError as e:
        print(f, "-", e.__class__.__name__)
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
import numpy as np
import aocl_compression_amd as a
import oracle_lib as ol
import bench
codec = sys.argv[1] if len(sys.argv) > 1 else "lz4"
cid, chunk = bench.CODECS[codec]
slot = (a.compress_bound(cid, chunk) + 15) & ~15
base = a.corpus.silesia_mix(64 << 20, seed=0xA0C1)
orc = ol.Oracle(); L = orc.L
ref = C.CDLL(ol.REF_SO)
cn, dn = bench.REF_FUNCS[codec]
fc = C.cast(getattr(ref, cn), C.c_void_p); fd = C.cast(getattr(ref, dn), C.c_void_p)
L.orc_bench_chunks.restype = C.c_int
L.orc_bench_chunks.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
for T in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    if T > (os.cpu_count() or 1):
        break
    n = min(2 << 30, (64 << 20) * T) // chunk * chunk
    sample = np.ascontiguousarray(np.tile(base, -(-n // len(base)))[:n])
    out_len = np.zeros(n // chunk, np.int32); tc = np.zeros(3); td = np.zeros(3)
    t0 = time.time()
    f = L.orc_bench_chunks(cid, fc, fd, sample.ctypes.data, n, chunk, slot, out_len.ctypes.data, T, 3, tc.ctypes.data, td.ctypes.data)
    g = n / 2**30
    print(json.dumps({"threads": T, "GiB": round(g, 3), "fails": f, "c_GiBps": round(g / tc.min(), 2), "d_GiBps": round(g / td.min(), 2), "wall_s": round(time.time() - t0, 1)}), flush=True)
