#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X block-codec engine.

Metric (BASELINE.json): GiB/s compress + decompress on a Silesia-shaped mix, 1/2/4/8 GPUs.
Headline workload = BASELINE config[1]: LZ4 level 1, 256 KiB independent chunks, the (synthetic) Silesia corpus
concatenated to 8 GiB per GPU, inputs resident in HBM before the timed region starts.  The same line also carries
BASELINE configs 3 and 4 under `codecs` (Snappy 64 KiB blocks on JSON/log records; Zstd level 1, 128 KiB blocks on the
Silesia mix), each driver-timed the same way with its own `roofline` and `cpu_baseline` (`--codec lz4|snappy|zstd` runs
one of them alone).  N > 1 = BASELINE config 5: LZ4 + Zstd, chunks sharded over the ranks, exact-size exchange of the
compacted compressed segments over RCCL.

One STEP of a codec = one pass of the hot path over the rank's shard:
    compress every chunk -> [N>1: compact + exchange of the compressed segments between all ranks] -> decompress.
`value` = uncompressed GiB that went through that round trip per second, summed over ranks (weak scaling: every rank
owns `--gib` GiB), from the max-over-ranks wall time of K steps bracketed by barrier + synchronize.

`roofline`: dominant kernel vs the 8 TB/s HBM peak; algorithmic bytes N + C per launch divided
by the live HIP-event time of that kernel on its own stream (aoclgpu_time_batch).  `cpu_baseline`: the real reference's
one-shot functions (oracle/_ref, kind "reference"; without that prebuilt library the key holds {"error": ...}) over the
same chunks on the host cores -- buffers first-touched by their threads, one warm-up pass, best / median of 5 passes,
one thread and all threads (oracle/chunk_driver.c orc_bench_chunks); rank 0, N=1 only.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GIB = float(1 << 30)
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SILESIA_BYTES = 211_938_580     # size of the real Silesia corpus; the synthetic base matches it
CODECS = {"lz4": (0, 262144), "snappy": (1, 65536), "zstd": (2, 131072), "zstd3": (2, 131072)}
ZSTD_LEVEL = {"zstd": 1, "zstd3": 3}            # zstd3 = BASELINE's Zstd chunk shape at level 3 ("double-fast")
REF_FUNCS = {"lz4": ("LZ4_compress_default", "LZ4_decompress_safe"), "snappy": ("snappy_compress", "snappy_uncompress"),
             "zstd": ("ZSTD_compress", "ZSTD_decompress"), "zstd3": ("ZSTD_compress", "ZSTD_decompress")}


# the sources a codec's kernels are built from: profiles/traffic.json records their hash next to the FETCH_SIZE / WRITE_SIZE
# passes, and a line whose kernels are not the profiled ones says so instead of quoting stale traffic
KERNEL_SOURCES = {"lz4": ["lz4_kernels.hip", "lz_group.hip.h", "common.hip.h"],
                  "snappy": ["snappy_kernels.hip", "lz_group.hip.h", "common.hip.h"],
                  "zstd": ["zstd_kernels.hip", "zstd_entropy.hip.h", "zstd_chain.h", "zstd_hufspec.h", "lz_group.hip.h", "common.hip.h"]}
KERNEL_SOURCES["zstd3"] = KERNEL_SOURCES["zstd"]


def kernel_source_sha16(codec):
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES[codec]:
        with open(os.path.join(ROOT, "aocl-compression_amd", "csrc", f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--codec", default="all", choices=["all", *CODECS],
                   help="all = lz4 (headline) + snappy + zstd + zstd3 (Zstd level 3) at N=1, lz4 + zstd at N>1")
    p.add_argument("--gib", type=float, default=8.0, help="uncompressed GiB per GPU and codec")
    p.add_argument("--chunk", type=int, default=0, help="chunk bytes (default: the codec's BASELINE chunk)")
    p.add_argument("--base-mib", type=float, default=SILESIA_BYTES / (1 << 20), help="size of the synthetic base corpus")
    p.add_argument("--cpu-sample-gib", type=float, default=2.0, help="cpu_baseline sample for the all-thread passes")
    p.add_argument("--cpu-reps", type=int, default=5)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-verify", action="store_true")
    p.add_argument("--no-by-kind", action="store_true", help="skip codecs.lz4.by_kind (kernel rates per corpus kind, 1 GiB each)")
    p.add_argument("--by-kind-all", action="store_true", help="by_kind for every codec of the run, not only lz4")
    p.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)
    p.add_argument("--exchange", default="c", choices=["c", "sendrecv", "allgather"],
                   help="N>1: the C library's exchange (aoclgpu_exchange_segments over its own RCCL communicator, default), "
                        "torch.distributed's grouped send/recv of exact sizes, or an all-gather of segments padded to the largest")
    return p.parse_args()


def self_launch(args):
    """`python3 bench.py --gpus N` (N > 1) started WITHOUT a launcher: this process starts the N ranks itself, as one child
    process (`python -m torch.distributed.run`, which spawns one worker per GPU), and only relays - the children's stdout and
    stderr are inherited, so rank 0's JSON line is this command's JSON line, and their exit code is this command's exit code.
    The parent never imports torch and never makes a HIP call (no exec of anything from a GPU-initialised process); the
    reference's caller does not launch workers by hand either (threads/threads.c:121-153, lz4.c:2684-2731)."""
    # (--standalone: the launcher picks a free rendezvous port itself and keeps it - a port found by bind-then-close here could be
    #  taken by another process before torch.distributed.run binds it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE"):          # a stale launcher variable without RANK: the launcher sets its own
        env.pop(k, None)
    env["AOCLGPU_BENCH_SELF_LAUNCHED"] = "1"
    if os.environ.get("AOCLGPU_BENCH_LAUNCH_DRYRUN") == "1":
        print(json.dumps({"launch": cmd, "torch_imported_by_parent": "torch" in sys.modules}), flush=True)
        return 0
    p = subprocess.run(cmd, env=env)
    assert "torch" not in sys.modules
    return p.returncode


def host_cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_budget():
    """(usable hardware threads, cgroup CPU quota or None): a container may see every hardware thread of the host and still
    be throttled to a few CPUs' worth of time per period (cgroup v2 cpu.max / v1 cfs_quota_us); OpenMP threads beyond that
    budget only add contention, so the all-thread baseline uses min(visible threads, quota) threads and reports both."""
    try:
        vis = len(os.sched_getaffinity(0))
    except AttributeError:
        vis = os.cpu_count() or 1
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(p)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    return vis, quota


def _median(xs):
    s = sorted(xs)
    return s[len(s) // 2] if len(s) % 2 else 0.5 * (s[len(s) // 2 - 1] + s[len(s) // 2])


def cpu_worker(spec):
    """Child process of cpu_baseline (started with OMP_PROC_BIND / OMP_PLACES in its environment, no GPU use):
    times the reference CPU path over the head of the chunk stream and prints one JSON object."""
    import numpy as np
    import oracle_lib as ol
    name, chunk, slot, reps = spec["codec"], spec["chunk"], spec["slot"], spec["reps"]
    base_np = np.load(spec["base"], mmap_mode="r")
    codec = CODECS[name][0]
    orc = ol.Oracle()
    cname, dname = REF_FUNCS[name]
    if not os.path.exists(ol.REF_SO):
        # cpu_baseline is the REAL reference or nothing: timing this repo's C restatement under the same key would silently
        # change what the number means
        print(json.dumps({"error": "oracle/_ref/libaocl_ref.so is not built (oracle/build_ref.sh needs /root/reference); no cpu_baseline"}))
        return
    ref = C.CDLL(ol.REF_SO)
    fc = C.cast(getattr(ref, cname), C.c_void_p)
    fd = C.cast(getattr(ref, dname), C.c_void_p)
    kind = "reference"
    L = orc.L
    if name in ZSTD_LEVEL:
        L.orc_set_zstd_level(ZSTD_LEVEL[name])
        if ref is not None:                            # the level table the reference's dispatcher installs (AOCL rows)
            ref.aocl_setup_zstd_encode.restype = C.c_void_p
            ref.aocl_setup_zstd_encode.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t]
            ref.aocl_setup_zstd_encode(0, 0, 0, 0, 0)
    L.orc_bench_chunks.restype = C.c_int
    L.orc_bench_chunks.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t,
                                   C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    cores = spec["cores"]

    def run(nbytes, threads):
        nbytes = min(nbytes, (64 << 20) * threads)
        n = int(min(spec["total"], nbytes)) // chunk * chunk
        r = -(-n // len(base_np))
        sample = np.ascontiguousarray(np.tile(base_np, r)[:n]) if r > 1 else np.ascontiguousarray(base_np[:n])
        out_len = np.zeros(n // chunk, np.int32)
        tc = np.zeros(reps, np.float64)
        td = np.zeros(reps, np.float64)
        fails = L.orc_bench_chunks(codec, fc, fd, sample.ctypes.data, n, chunk, slot, out_len.ctypes.data, threads, reps,
                                   tc.ctypes.data, td.ctypes.data)
        assert fails == 0, f"cpu_baseline: {fails} failures"
        g = n / GIB
        return {
            "threads": threads, "sample_GiB": round(g, 4),
            "compress_GiBps": round(g / tc.min(), 4), "decompress_GiBps": round(g / td.min(), 4),
            "compress_GiBps_median": round(g / _median(tc.tolist()), 4),
            "decompress_GiBps_median": round(g / _median(td.tolist()), 4),
            "value": round(g / (tc.min() + td.min()), 4),
        }, out_len

    one, _ = run(64 << 20, 1)
    allt, out_len = run(spec["sample_gib"] * GIB, cores)
    np.save(spec["lens_out"], out_len)
    print(json.dumps({
        "value": allt["value"], "unit": "GiB/s", "cores": cores, "kind": kind,
        "compress_GiBps": allt["compress_GiBps"], "decompress_GiBps": allt["decompress_GiBps"],
        "all_threads": allt, "one_thread": one,
        "one_thread_MBps": {"compress": round(one["compress_GiBps"] * GIB / 1e6, 1), "decompress": round(one["decompress_GiBps"] * GIB / 1e6, 1)},
        "host_cpu": host_cpu_model(), "host_threads_visible": spec["visible"], "cgroup_cpu_quota": spec["quota"],
        "omp": {k: os.environ.get(k) for k in ("OMP_PROC_BIND", "OMP_PLACES")},
        "method": f"child process; buffers first-touched by their threads, 1 warm-up pass, best of {reps} passes (median "
                  "alongside), schedule(static,1) over chunks (oracle/chunk_driver.c orc_bench_chunks)",
        "sample": f"first {allt['sample_GiB']:.2f} GiB (all threads) / {one['sample_GiB'] * 1024:.0f} MiB (1 thread) of the same "
                  f"{chunk // 1024} KiB chunk stream, {cname}" + (f"(level {ZSTD_LEVEL[name]})" if name in ZSTD_LEVEL else "") + f" + {dname}",
    }), flush=True)


_BASE_FILES = {}


def cpu_baseline(codec_name, base_np, total, chunk, slot, sample_gib, reps):
    """The reference CPU path over the head of the same chunk stream, 1 thread and all threads, timed in a child process
    so that the OpenMP binding (OMP_PROC_BIND=close, one place per hardware thread) cannot pin this process's HIP runtime
    threads.  Returns (record, per-chunk compressed sizes of the all-thread sample)."""
    import numpy as np
    visible, quota = cpu_budget()
    cores = visible if quota is None else max(1, min(visible, int(quota + 0.5)))
    tmp = tempfile.gettempdir()
    key = id(base_np)
    if key not in _BASE_FILES:
        path = os.path.join(tmp, f"aoclgpu_bench_base_{os.getpid()}_{len(_BASE_FILES)}.npy")
        np.save(path, base_np)
        _BASE_FILES[key] = path
    lens_out = os.path.join(tmp, f"aoclgpu_bench_lens_{os.getpid()}.npy")
    spec = {"codec": codec_name, "chunk": chunk, "slot": slot, "reps": reps, "total": total, "sample_gib": sample_gib,
            "cores": cores, "visible": visible, "quota": quota, "base": _BASE_FILES[key], "lens_out": lens_out}
    env = dict(os.environ)
    env.setdefault("OMP_PROC_BIND", "close")
    env.setdefault("OMP_PLACES", "threads")
    env["HIP_VISIBLE_DEVICES"] = ""          # the child never touches the GPU
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-worker", json.dumps(spec)], env=env,
                       capture_output=True, text=True, timeout=600)
    if p.returncode != 0:
        raise RuntimeError("cpu_baseline worker failed: " + p.stderr[-400:])
    rec = json.loads(p.stdout.strip().splitlines()[-1])
    if "error" in rec:
        raise RuntimeError(rec["error"])
    out_len = np.load(lens_out)
    os.unlink(lens_out)
    return rec, out_len


def whole_input_frames(eng):
    """Zstd inputs above 128 KiB are frames of several blocks: one wavefront per frame.  Reported next to
    the headline, never part of it: 512 frames of 1 MiB side by side (data already in HBM), one 1 MiB frame alone, and the
    same two for an LZ4 frame with linked blocks through LZ4F_compressFrame (host buffers).  A sample of the frames is
    compared with the reference's ZSTD_compress / LZ4F_compressFrame when the reference build is present."""
    import aocl_compression_amd as a
    import numpy as np
    import torch
    import oracle_lib as ol
    out = {}
    fb, k = 1 << 20, 512
    base = a.corpus.silesia_mix(64 << 20, seed=0x5EED)
    src = torch.from_numpy(base).to(eng.device).repeat(k * fb // base.size + 1)[:k * fb].contiguous()
    off = np.arange(k + 1, dtype=np.uint64) * fb
    w, h, m = 20, 14, 7                                        # ZSTD_getCParams(1, 1 MiB, 0) under the AOCL level table
    best = None
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        dst, doff, lens = eng.zstd_compress_frames(src, off, w, h, m)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = dt if best is None or (it and dt < best) else best
    assert all(x > 0 for x in lens)
    out["zstd_level1_frames_1MiB_x512"] = {"GiBps": round(k * fb / best / 2**30, 2), "ms": round(best * 1e3, 1), "ratio": round(k * fb / sum(lens), 3)}
    one = src[:fb].contiguous()
    best = None
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        d1, do1, l1 = eng.zstd_compress_frames(one, [0, fb], w, h, m)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = dt if best is None or (it and dt < best) else best
    out["zstd_level1_one_frame_1MiB"] = {"MiBps": round(fb / best / 2**20, 1), "ms": round(best * 1e3, 1)}
    if os.path.exists(ol.REF_SO):
        ref = C.CDLL(ol.REF_SO)
        ref.ZSTD_compress.restype = C.c_size_t
        ref.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
        host = dst.cpu().numpy()
        same = True
        for i in (0, 1, k // 2, k - 1):
            data = src[i * fb:(i + 1) * fb].cpu().numpy().tobytes()
            cap = fb + (fb >> 8) + 64
            ob = C.create_string_buffer(cap)
            r = ref.ZSTD_compress(ob, cap, data, fb, 1)
            same = same and host[doff[i]:doff[i] + lens[i]].tobytes() == ob.raw[:r]
        out["zstd_frames_identical_to_reference_ZSTD_compress"] = bool(same)
        # the reference on ONE host core for the same frames (cpu_baseline of this report; best of 3 passes over 32 frames)
        datas = [src[i * fb:(i + 1) * fb].cpu().numpy().tobytes() for i in range(32)]
        ob = C.create_string_buffer(fb + (fb >> 8) + 64)
        bestc = None
        for _ in range(3):
            t0 = time.perf_counter()
            for d_ in datas:
                ref.ZSTD_compress(ob, len(ob), d_, fb, 1)
            dt = time.perf_counter() - t0
            bestc = dt if bestc is None or dt < bestc else bestc
        out["cpu_baseline"] = {"kind": "reference", "cores": 1, "unit": "MB/s", "value": round(32 * fb / bestc / 1e6, 1),
                               "ms_per_1MiB_frame": round(bestc / 32 * 1e3, 3), "sample": "32 frames of 1 MiB, ZSTD_compress(.., 1), best of 3"}
    for key, ms_key, nb in (("zstd_level1_frames_1MiB_x512", "ms", k * fb), ("zstd_level1_one_frame_1MiB", "ms", fb)):
        if key in out:                                     # roofline of the frame kernels: N + C bytes per launch over the wall time
            rec = out[key]
            comp = nb / rec["ratio"] if "ratio" in rec else nb / out["zstd_level1_frames_1MiB_x512"]["ratio"]
            ach = (nb + comp) / (rec[ms_key] * 1e-3) / 1e9
            rec["roofline"] = {"bound": "hbm", "kernel": "zstd_frame_kernel" if key.endswith("x512") else "zstd_frame_lds_kernel",
                               "achieved": round(ach, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 7)}
    try:                                                       # LZ4 frame with linked blocks (the format's default), host buffers
        import lz4f_lib as lf
        G = lf.bind(eng.lib)
        data = base.tobytes()[:4 << 20]
        p = lf.prefs()
        t0 = time.perf_counter(); frame = lf.compress(G, data, p); dt = time.perf_counter() - t0
        t0 = time.perf_counter(); frame = lf.compress(G, data, p); dt = min(dt, time.perf_counter() - t0)
        res = {"MiBps": round(len(data) / dt / 2**20, 1), "ms": round(dt * 1e3, 1), "ratio": round(len(data) / len(frame), 3)}
        if os.path.exists(ol.REF_SO) and hasattr(C.CDLL(ol.REF_SO), "LZ4F_compressFrame"):
            R = lf.bind(C.CDLL(ol.REF_SO))
            t0 = time.perf_counter(); rf = lf.compress(R, data, p); dtr = time.perf_counter() - t0
            t0 = time.perf_counter(); rf = lf.compress(R, data, p); dtr = min(dtr, time.perf_counter() - t0)
            res["identical_to_reference_LZ4F_compressFrame"] = bool(frame == rf)
            res["cpu_baseline"] = {"kind": "reference", "cores": 1, "unit": "MB/s", "value": round(len(data) / dtr / 1e6, 1), "ms": round(dtr * 1e3, 3)}
        out["lz4f_linked_one_frame_4MiB_host"] = res
    except Exception as e:
        out["lz4f_linked_one_frame_4MiB_host"] = {"error": repr(e)}
    out["note"] = "one wavefront per frame (blocks of a frame depend on each other in order); never the reported value"
    return out


def memory_probes(eng):
    """The memory system's own numbers on this box, next to the data-sheet peak the roofline is priced against:
    a device-to-device copy (coalesced 16-byte lanes, 2 GiB) and a scattered 8-byte-load kernel (the codecs' candidate-window
    pattern; profiles/ holds the FETCH_SIZE pass of this same kernel = bytes the counter reports per scattered load)."""
    import torch
    n = 2 << 30
    a_ = torch.empty(n, dtype=torch.uint8, device=eng.device)
    b_ = torch.ones(n, dtype=torch.uint8, device=eng.device)
    ms = eng.lib.aoclgpu_probe_copy(a_.data_ptr(), b_.data_ptr(), n, 5, eng._stream())
    out = {"hbm_copy_GBps": round(2.0 * n / (ms * 1e-3) / 1e9, 1) if ms > 0 else None, "hbm_copy_bytes": n}
    try:
        eng.lib.aoclgpu_probe_copy_variant.restype = C.c_char_p
        out["hbm_copy_variant"] = eng.lib.aoclgpu_probe_copy_variant().decode()
    except AttributeError:
        pass
    tot = C.c_uint64(0)
    ms2 = eng.lib.aoclgpu_probe_scatter8(b_.data_ptr(), n, 256, C.byref(tot), eng._stream())
    if ms2 > 0:
        cal = {"kernel": "probe_scatter8_kernel", "loads": tot.value, "requested_bytes": 8 * tot.value, "ms": round(ms2, 3),
               "Gloads_per_s": round(tot.value / (ms2 * 1e-3) / 1e9, 2)}
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                t = json.load(f).get("scatter8_calibration")
            if t:
                cal["fetch_size_bytes_per_load_measured"] = t["fetch_bytes_per_load"]
                cal["source"] = t["source"]
        except (OSError, ValueError, KeyError):
            pass
        out["fetch_calibration"] = cal
    del a_, b_
    torch.cuda.empty_cache()
    return out


def host_path(eng, gib=1.0):
    """The drop-in as a reference user calls it: 1 GiB of the LZ4 bench stream in PAGEABLE host memory through aocl_llc_compress /
    aocl_llc_decompress with numThreads = 2 (the RAP stream path: partitions on the GPU, PCIe both ways, staging and assembly
    included).  Outside the timed region, rank 0, N = 1; never the reported value."""
    import numpy as np
    import aocl_compression_amd as a
    n = int(gib * (1 << 30))
    data = a.corpus.silesia_mix(min(n, 256 << 20), seed=0xA0C1)
    src = np.ascontiguousarray(np.tile(data, -(-n // data.size))[:n])
    lib = eng.lib
    lib.aocl_get_rap_frame_bound_mt.restype = C.c_int
    cap = a.compress_bound(a.LZ4, n) + int(lib.aocl_get_rap_frame_bound_mt()) + (n >> 8)
    out = np.empty(cap, dtype=np.uint8)
    back = np.empty(n, dtype=np.uint8)
    d = a.api.LlcDesc(inBuf=src.ctypes.data, outBuf=out.ctypes.data, inSize=n, outSize=cap, level=1, numThreads=2)
    res = {"bytes": n, "numThreads": 2, "buffers": "pageable host memory"}
    best = None
    for _ in range(3):
        d.inBuf, d.outBuf, d.inSize, d.outSize = src.ctypes.data, out.ctypes.data, n, cap
        t0 = time.perf_counter()
        r = lib.aocl_llc_compress(C.byref(d), a.api._LLC_ENUM[a.LZ4])
        dt = time.perf_counter() - t0
        if r <= 0:
            return {"error": f"aocl_llc_compress returned {r}"}
        best = dt if best is None else min(best, dt)
    res["compress_GiBps"] = round(n / (1 << 30) / best, 2); res["compress_ms"] = round(best * 1e3, 1); res["compressed_bytes"] = int(r)
    dd = a.api.LlcDesc(inBuf=out.ctypes.data, outBuf=back.ctypes.data, inSize=int(r), outSize=n, numThreads=2)
    best = None
    for _ in range(3):
        dd.inBuf, dd.outBuf, dd.inSize, dd.outSize = out.ctypes.data, back.ctypes.data, int(r), n
        t0 = time.perf_counter()
        rr = lib.aocl_llc_decompress(C.byref(dd), a.api._LLC_ENUM[a.LZ4])
        dt = time.perf_counter() - t0
        if rr != n:
            return {"error": f"aocl_llc_decompress returned {rr}"}
        best = dt if best is None else min(best, dt)
    res["decompress_GiBps"] = round(n / (1 << 30) / best, 2); res["decompress_ms"] = round(best * 1e3, 1)
    res["round_trip"] = bool(np.array_equal(back, src))
    return res


def by_kind(eng, name, gib=1.0, base_mib=32):
    """Kernel rates of one codec per corpus kind (the Silesia-shaped mix is 43 % text, 26 % binary, 12 % JSON/log, 8 % database,
    9 % 16-bit image, 2 % random): 1 GiB of each kind alone, HBM-resident, compress and decompress launches timed with HIP events
    (best of 2).  A regression on one kind is visible here before it moves the mix.  Outside the timed region, rank 0, N = 1."""
    import torch
    import aocl_compression_amd as a
    codec, chunk = CODECS[name]
    out = {}
    total = int(gib * GIB) // chunk * chunk
    n = total // chunk
    slot = (a.compress_bound(codec, chunk) + 15) & ~15
    slots = torch.empty(n * slot, dtype=torch.uint8, device=eng.device)
    clen = torch.zeros(n, dtype=torch.int32, device=eng.device)
    dst = torch.empty(total, dtype=torch.uint8, device=eng.device)
    dlen = torch.zeros(n, dtype=torch.int32, device=eng.device)
    for kind in a.corpus.KINDS:
        base = torch.from_numpy(a.corpus.make(kind, int(base_mib) << 20, seed=7)).to(eng.device)
        src = base.repeat(-(-total // base.numel()))[:total].contiguous()
        cb = eng.uniform_compress_batch(src, chunk, slots, clen, slot)
        if ZSTD_LEVEL.get(name, 1) != 1:
            cb.reserved |= a.api.zstd_level(ZSTD_LEVEL[name])
        db = eng.uniform_decompress_batch(slots, clen, slot, dst, chunk, dlen)
        eng.time_batch(codec, 0, cb, 1)
        tc = min(eng.time_batch(codec, 0, cb, 1) for _ in range(2))
        eng.time_batch(codec, 1, db, 1)
        td = min(eng.time_batch(codec, 1, db, 1) for _ in range(2))
        ok = bool(torch.equal(dst, src))
        out[kind] = {"ratio": round(total / max(int(clen.sum().item()), 1), 3), "compress_GiBps": round(total / GIB / (tc * 1e-3), 1),
                     "decompress_GiBps": round(total / GIB / (td * 1e-3), 1), "round_trip": ok}
        del src, base
    out["note"] = f"{gib:g} GiB per kind ({base_mib} MiB base, tiled), {chunk // 1024} KiB chunks, kernel launches only (HIP events, best of 2)"
    del slots, dst
    torch.cuda.empty_cache()
    return out


def two_batches_in_flight(eng, args, base_np, steps=6):
    """LZ4, N = 1, outside the timed region and NOT `value`: the bench's step (compress the 8 GiB batch, then decompress it) with TWO
    batches in flight - consecutive steps alternate between two HIP streams and two sets of output buffers, so that the thin end of
    one kernel (a launch ends at ~1.1x, the decoder's at ~1.2x, of the time its chunks would take on a chip that stayed full:
    tools/chunk_time_study.py) runs beside the next batch's kernels.  What a service that always has a next batch gains over the
    one-batch-at-a-time number the headline is; both are measured here back to back on the same buffers."""
    import torch
    import aocl_compression_amd as a
    codec, chunk = CODECS["lz4"]
    chunk = args.chunk or chunk
    total = int(args.gib * GIB) // chunk * chunk
    n = total // chunk
    slot = (a.compress_bound(codec, chunk) + 15) & ~15
    dev = eng.device
    base = torch.from_numpy(base_np).to(dev)
    src = base.repeat(-(-total // base.numel()))[:total].contiguous()
    del base
    bufs, cb, db = [], [], []
    for _ in range(2):
        slots = torch.empty(n * slot, dtype=torch.uint8, device=dev)
        clen = torch.zeros(n, dtype=torch.int32, device=dev)
        dst = torch.empty(total, dtype=torch.uint8, device=dev)
        dlen = torch.zeros(n, dtype=torch.int32, device=dev)
        bufs.append((slots, clen, dst, dlen))
        cb.append(eng.uniform_compress_batch(src, chunk, slots, clen, slot))
        db.append(eng.uniform_decompress_batch(slots, clen, slot, dst, chunk, dlen))
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    fc, fd = eng._fn(codec, "compress"), eng._fn(codec, "decompress")

    def run(k_steps, two):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(k_steps):
            i = (k & 1) if two else 0
            with torch.cuda.stream(streams[i]):
                eng._check(fc(C.byref(cb[i]), eng._stream()), "compress")
                eng._check(fd(C.byref(db[i]), eng._stream()), "decompress")
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k_steps * 1e3

    run(2, True)
    one = min(run(steps, False) for _ in range(2))
    two = min(run(steps, True) for _ in range(2))
    ok = all(bool(torch.equal(b[2], src)) and int((b[1] <= 0).sum().item()) == 0 for b in bufs)
    gib = total / GIB
    return {"codec": "lz4", "steps": steps, "gib_per_step": round(gib, 3),
            "one_stream_ms_per_step": round(one, 3), "one_stream_GiBps": round(gib / (one / 1e3), 3),
            "two_streams_ms_per_step": round(two, 3), "two_streams_GiBps": round(gib / (two / 1e3), 3),
            "round_trip": ok,
            "note": "same kernels, same bytes per step; steps alternate between two HIP streams and two sets of slots / output buffers "
                    "(wall clock over the steps, no per-kernel events: the kernels of the two batches overlap).  Not `value`: the headline times one batch at a time"}


def config1_latency(eng, reps=20):
    """BASELINE config 1: one 64 KiB text block through aocl_llc_compress / aocl_llc_decompress on HOST buffers
    (PCIe both ways included), next to the reference's own aocl_llc_* on one host core.  Microseconds, best of `reps`."""
    import aocl_compression_amd as a
    import oracle_lib as ol
    data = a.corpus.make("text", 65536, seed=0xA0C1 + 1).tobytes()
    cap = a.compress_bound(a.LZ4, len(data))
    out = {}

    def best(fn):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return min(ts) * 1e6

    comp = {}
    def gc():
        comp["r"] = eng.llc_compress(a.LZ4, data, cap)
    out["gpu_compress_us"] = round(best(gc), 1)
    r, blob, _ = comp["r"]
    assert r > 0
    def gd():
        rr, back, _ = eng.llc_decompress(a.LZ4, blob, len(data))
        assert rr == len(data)
    out["gpu_decompress_us"] = round(best(gd), 1)
    if os.path.exists(ol.REF_SO):
        ref = C.CDLL(ol.REF_SO)
        ref.LZ4_compress_default.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
        ref.LZ4_decompress_safe.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
        ob = C.create_string_buffer(cap)
        bb = C.create_string_buffer(len(data))
        n = {}
        def rc():
            n["c"] = ref.LZ4_compress_default(data, ob, len(data), cap)
        out["reference_1core_compress_us"] = round(best(rc), 1)
        assert ob.raw[:n["c"]] == blob, "config 1: GPU bytes differ from the reference's"
        def rd():
            ref.LZ4_decompress_safe(ob, bb, n["c"], len(data))
        out["reference_1core_decompress_us"] = round(best(rd), 1)
        out["bytes_identical_to_reference"] = True
    out["note"] = "host buffers in and out (H2D + one wavefront + D2H per call); never the reported value"
    return out


class Run:
    """Everything one codec needs on this rank: the HBM-resident shard, slots, batches and the exchange buffers."""

    def __init__(self, eng, dist, args, name, base_np, corpus_name, rank, world):
        import torch
        import aocl_compression_amd as a
        self.torch, self.eng, self.dist, self.args, self.name = torch, eng, dist, args, name
        self.rank, self.world = rank, world
        self.codec, def_chunk = CODECS[name]
        self.chunk = chunk = args.chunk or def_chunk
        self.total = total = int(args.gib * GIB) // chunk * chunk
        self.nchunks = nchunks = total // chunk
        self.slot = slot = (a.compress_bound(self.codec, chunk) + 15) & ~15
        self.corpus_name = corpus_name
        dev = eng.device
        self.base = torch.from_numpy(base_np).to(dev)
        self.src = self.shard_of(rank)
        self.slots = torch.empty(nchunks * slot, dtype=torch.uint8, device=dev)
        self.clen = torch.zeros(nchunks, dtype=torch.int32, device=dev)
        self.dst = torch.empty(total, dtype=torch.uint8, device=dev)
        self.dlen = torch.zeros(nchunks, dtype=torch.int32, device=dev)
        self.aux = torch.zeros(nchunks, dtype=torch.int32, device=dev)     # LZ4: final literal run per chunk (RAP join)
        self.cb = eng.uniform_compress_batch(self.src, chunk, self.slots, self.clen, slot)
        self.cb.aux = self.aux.data_ptr()
        if ZSTD_LEVEL.get(name, 1) != 1:
            self.cb.reserved |= a.api.zstd_level(ZSTD_LEVEL[name])
        self.db = eng.uniform_decompress_batch(self.slots, self.clen, slot, self.dst, chunk, self.dlen)
        self.kt = {"c": [], "d": []}
        self.df_slabs = 0
        # what the COMMAND asked for (captured once in main, before any codec's run could change args.exchange), and every change of
        # the exchange away from it so far - a fallback an earlier codec of this run made is this codec's fallback too
        self.exchange_requested = getattr(args, "exchange_requested", args.exchange)
        if not hasattr(args, "exchange_fallbacks"):
            args.exchange_fallbacks = []
        self.fallbacks = args.exchange_fallbacks
        if world > 1:
            self.setup_exchange()

    def shard_of(self, r):
        """Rank r's shard of the stream: the base corpus rotated by a rank-specific offset, tiled to `total` bytes."""
        torch = self.torch
        shift = (r * 7919 * 4096) % self.base.numel()
        b = torch.roll(self.base, -shift) if shift else self.base
        reps = -(-self.total // b.numel())
        return b.repeat(reps)[:self.total].contiguous()

    # ---- N > 1: exchange of the compacted compressed segments ----
    def all_gather_vec(self, out, mine):
        """all-gather of one device vector per rank (gloo, the 1-GPU validation backend, only moves host tensors)"""
        if getattr(self, "cx", None) and mine.dtype == self.torch.int32:
            return self.cx.allgather_words(out, mine)
        if not self.gloo:
            return self.dist.all_gather_into_tensor(out, mine)
        parts = [self.torch.zeros(mine.numel(), dtype=mine.dtype) for _ in range(self.world)]
        self.dist.all_gather(parts, mine.cpu())
        out.copy_(self.torch.cat(parts))

    def setup_exchange(self):
        torch, dist, eng = self.torch, self.dist, self.eng
        dev = eng.device
        self.gloo = dist.get_backend() == "gloo"
        # segment sizes are deterministic: one untimed compress pass fixes them for every rank
        eng.time_batch(self.codec, 0, self.cb, 1)
        mine = self.clen.sum().to(torch.int64).reshape(1)
        sizes = torch.zeros(self.world, dtype=torch.int64, device=dev)
        self.all_gather_vec(sizes, mine)
        self.seg = [int(x) for x in sizes.cpu().tolist()]
        self.off = torch.zeros(self.nchunks + 1, dtype=torch.int64, device=dev)
        self.len_all = torch.zeros(self.world * self.nchunks, dtype=torch.int32, device=dev)
        self.aux_all = torch.zeros(self.world * self.nchunks, dtype=torch.int32, device=dev)
        self.all_gather_vec(self.aux_all, self.aux)
        if self.args.exchange == "c" and self.gloo:
            self.args.exchange = "sendrecv"           # host-staged validation runs (1 GPU / CPU) have no RCCL communicator
            self.fallbacks.append("c -> sendrecv: torch.distributed backend is gloo (host-staged validation run, no RCCL communicator)")
        if self.args.exchange == "c" and not getattr(self, "cx", None):
            # the product path: the communicator and the exchange are the C library's (csrc/exchange.cpp).  Every rank learns
            # together whether it came up; if not, all of them use torch.distributed's grouped send / recv instead.
            from aocl_compression_amd import sharding
            ok = 1
            try:
                self.cx = sharding.CExchange(eng, dist, self.rank, self.world)
            except Exception as e:  # noqa: BLE001
                ok = 0
                why = repr(e)
                if self.rank == 0:
                    print(f"bench: C exchange unavailable ({e!r}); using torch.distributed send/recv", file=sys.stderr)
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                self.cx = None
                self.args.exchange = "sendrecv"
                self.fallbacks.append("c -> sendrecv: the library's RCCL communicator did not come up on every rank"
                                      + (f" (this rank: {why})" if not ok else " (another rank)"))
        if self.args.exchange == "allgather":
            # RCCL has no all-gatherv: every segment is padded to the largest one (the ring moves the padding too)
            self.seg_cap = (max(self.seg) + (1 << 20)) & ~0xFFFFF
            self.seg_off = [r * self.seg_cap for r in range(self.world)]
            self.packed = torch.empty(self.seg_cap, dtype=torch.uint8, device=dev)
            self.gather_buf = torch.empty(self.world * self.seg_cap, dtype=torch.uint8, device=dev)
        else:
            # exact sizes (aocl_compression_amd.sharding): rank r's segment lives at the exclusive prefix sum of the segment
            # sizes (16-byte aligned starts); this rank compacts straight into its own place
            from aocl_compression_amd import sharding
            self.seg_off, total = sharding.segment_layout(self.seg)
            self.gather_buf = torch.empty(max(total, 16), dtype=torch.uint8, device=dev)
            self.packed = self.gather_buf[self.seg_off[self.rank]:self.seg_off[self.rank] + self.seg[self.rank]]
            # one untimed exchange: if this RCCL build cannot run the grouped send/recv, every rank sees the same exception and
            # all of them fall back to the padded all-gather together (the JSON line says which exchange ran)
            # ... and one that never returns (two communicators on one device, a link that does not come up) must not take the
            # whole run with it: a watchdog ends this process with a JSON error line after 60 s - no retry, no re-exec
            import threading

            def _expired():
                print(json.dumps({"error": "exchange timeout", "exchange": self.args.exchange, "rank": self.rank, "n_gpus": self.world,
                                  "note": "the first (untimed) exchange of compressed segments did not complete within 60 s"}), flush=True)
                os._exit(3)
            dog = threading.Timer(60.0, _expired)
            dog.daemon = True
            dog.start()
            try:
                for w in self.exchange():
                    w.wait()
                done = torch.cuda.Event()
                done.record()
                while not done.query():
                    time.sleep(0.005)
                dog.cancel()
            except Exception as e:  # noqa: BLE001
                dog.cancel()
                nxt = "sendrecv" if self.args.exchange == "c" else "allgather"
                if self.rank == 0:
                    print(f"bench: {self.args.exchange} exchange failed ({e!r}); using {nxt}", file=sys.stderr)
                self.fallbacks.append(f"{self.args.exchange} -> {nxt}: the first (untimed) exchange raised {e!r}")
                self.args.exchange = nxt
                self.cx = None
                self.setup_exchange()

    def exchange(self):
        dist, torch = self.dist, self.torch
        if self.args.exchange == "allgather":
            if self.gloo:                             # validation on one GPU: same data movement, staged through the host
                parts = [torch.zeros(self.seg_cap, dtype=torch.uint8) for _ in range(self.world)]
                dist.all_gather(parts, self.packed.cpu())
                self.gather_buf.copy_(torch.cat(parts))
                return []
            return [dist.all_gather_into_tensor(self.gather_buf, self.packed, async_op=True)]
        if self.args.exchange == "c":
            return self.cx.exchange(self.gather_buf, self.seg, self.seg_off)
        from aocl_compression_amd import sharding
        return sharding.exchange_segments(dist, self.rank, self.world, self.gather_buf, self.seg, self.seg_off, host_staged=self.gloo)

    def step(self, record):
        eng = self.eng
        tc = eng.time_batch(self.codec, 0, self.cb, 1)          # HIP events around the compress kernel, same stream
        if self.name == "zstd3":                                # read right behind the call, on the calling thread (a process-wide value)
            self.df_slabs = int(eng.lib.aoclgpu_zstd_last_split_slabs())
        works = []
        if self.world > 1:
            rc = eng.lib.aoclgpu_compact_slots(self.slots.data_ptr(), self.slot, self.clen.data_ptr(), self.nchunks,
                                               self.packed.data_ptr(), self.off.data_ptr(), eng._stream())
            assert rc == 0
            self.all_gather_vec(self.len_all, self.clen)
            # the exchange runs on RCCL's stream while this rank decodes its own chunks
            works = self.exchange()
        td = eng.time_batch(self.codec, 1, self.db, 1)
        for w in works:
            w.wait()
        if record:
            self.kt["c"].append(tc)
            self.kt["d"].append(td)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def timed(self):
        args, torch, dist = self.args, self.torch, self.dist
        for _ in range(args.warmup):
            self.step(False)
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            self.step(True)
        self.barrier()
        elapsed = time.perf_counter() - t0
        if self.world > 1:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=self.eng.device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        return elapsed

    def exchange_report(self, ms_step, ms_c, ms_d):
        """N > 1, outside the timed region, every rank: the exchange ALONE (no kernel beside it) - barrier, one exchange of the
        compacted segments as the step issues it, wait, synchronize; max over ranks, best of 3 - so that a scaling shortfall can
        be split into kernels and links.  `exchange_GBps` = bytes one GPU RECEIVES (all other ranks' segments) over that time,
        next to what its links carry inbound ((N-1) peers x ~153 GB/s xGMI, MI355X_MICROARCH.md); `exchange_exposed_ms` = step
        time minus the two kernels (HIP events, max over ranks) = what the exchange, the compaction and the size gather cost
        the step after overlapping with decompress.  `rccl_nranks` = ncclCommCount of the library's communicator."""
        torch, dist = self.torch, self.dist
        best = None
        for _ in range(3):
            self.barrier()
            t0 = time.perf_counter()
            for w in self.exchange():
                w.wait()
            torch.cuda.synchronize()
            tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=self.eng.device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            best = float(tt.item()) if best is None else min(best, float(tt.item()))
        total = float(sum(self.seg))
        if self.args.exchange == "allgather":
            recv = float(self.seg_cap) * (self.world - 1)             # the ring moves every padded segment to every rank
        else:
            recv = (total - min(self.seg)) if self.world > 1 else 0.0  # the rank with the smallest own segment receives the most
        cx = getattr(self, "cx", None)
        links = min(self.world - 1, 7)
        return {
            "exchange": {"c": "aoclgpu_exchange_segments (C library, own RCCL communicator, grouped exact-size send/recv)",
                         "sendrecv": "torch.distributed batch_isend_irecv of exact sizes" + (" (host-staged over gloo)" if self.gloo else ""),
                         "allgather": "all-gather of segments padded to the largest" + (" (host-staged over gloo)" if self.gloo else "")}[self.args.exchange],
            "exchange_requested": self.exchange_requested,
            "exchange_fell_back": bool(self.fallbacks), "exchange_fallback_reason": "; ".join(self.fallbacks) or None,
            "rccl_nranks": cx.nranks() if cx else None,
            "torch_distributed_backend": dist.get_backend(),
            "exchange_ms": round(best * 1e3, 3), "exchange_recv_bytes_per_gpu": recv,
            "exchange_GBps": round(recv / best / 1e9, 2) if best > 0 else None,
            "exchange_GBps_all_gpus": round(recv * self.world / best / 1e9, 2) if best > 0 else None,
            "xgmi_inbound_peak_GBps_per_gpu": 153.0 * links,
            "exchange_frac_of_xgmi_peak": round(recv / best / 1e9 / (153.0 * links), 4) if best > 0 and links else None,
            "exchange_exposed_ms": round(ms_step - ms_c - ms_d, 3),
            "exchange_note": "exchange_ms: one exchange alone (nothing overlapped), max over ranks, best of 3; exchange_exposed_ms: "
                             "step - compress kernel - decompress kernel (includes compaction and the per-chunk size gather)",
        }

    # ---- checks outside the timed region ----
    def verify_local(self):
        torch = self.torch
        assert int((self.clen <= 0).sum().item()) == 0, "a chunk failed to compress"
        assert int((self.dlen != self.chunk).sum().item()) == 0, "a chunk failed to decompress"
        assert torch.equal(self.dst, self.src), "round trip mismatch"

    def verify_remote(self):
        """Decode the segment this rank RECEIVED from the next rank (sizes from the gathered length table) and compare
        it with that rank's shard, regenerated here: a wrong exchange cannot print a number."""
        import aocl_compression_amd as a
        torch, eng = self.torch, self.eng
        peer = (self.rank + 1) % self.world
        lens = self.len_all[peer * self.nchunks:(peer + 1) * self.nchunks]
        assert int(lens.sum().item()) == self.seg[peer], "gathered chunk sizes do not add up to the peer's segment size"
        csum = torch.cumsum(lens.to(torch.int64), 0)
        src_off = (csum - lens + self.seg_off[peer]).contiguous()
        dst_off = (torch.arange(self.nchunks, dtype=torch.int64, device=eng.device) * self.chunk).contiguous()
        dst_cap = torch.full((self.nchunks,), self.chunk, dtype=torch.int32, device=eng.device)
        out = torch.empty(self.total, dtype=torch.uint8, device=eng.device)
        olen = torch.zeros(self.nchunks, dtype=torch.int32, device=eng.device)
        b = a.Batch(src=self.gather_buf.data_ptr(), dst=out.data_ptr(), src_off=src_off.data_ptr(),
                    src_len=lens.contiguous().data_ptr(), dst_off=dst_off.data_ptr(), dst_cap=dst_cap.data_ptr(),
                    out_len=olen.data_ptr(), nchunks=self.nchunks)
        eng._check(eng._fn(self.codec, "decompress")(C.byref(b), eng._stream()), "decode of the gathered segment")
        torch.cuda.synchronize()
        assert int((olen != self.chunk).sum().item()) == 0, f"rank {self.rank}: a chunk of rank {peer}'s gathered segment failed to decode"
        want = self.shard_of(peer)
        assert torch.equal(out, want), f"rank {self.rank}: rank {peer}'s gathered segment does not decode to its shard"
        return peer

    def verify_assemble(self, slab_bytes=1 << 30):
        """Config 5 as a product: ONE RAP stream (threads/threads.h:77-106) from the gathered segments of TWO ranks - the last
        `slab_bytes`/2 of this rank's shard and the first `slab_bytes`/2 of the next rank's - assembled by
        aoclgpu_rap_assemble from the gathered buffer and size tables alone, then decoded by aoclgpu_rap_unpack and compared
        with the source bytes of both ranks."""
        import numpy as np
        torch, eng = self.torch, self.eng
        peer = (self.rank + 1) % self.world
        half = max(1, min(self.nchunks, slab_bytes // 2 // self.chunk, 32767))
        lens = self.len_all.cpu().numpy().reshape(self.world, self.nchunks)
        auxs = self.aux_all.cpu().numpy().reshape(self.world, self.nchunks)
        offs = []
        for r in range(self.world):
            c = np.cumsum(lens[r].astype(np.int64))
            offs.append(c - lens[r] + self.seg_off[r])
        pick = [(self.rank, self.nchunks - half, self.nchunks), (peer, 0, half)]
        off = np.concatenate([offs[r][a:b] for r, a, b in pick]).astype(np.uint64)
        ln = np.concatenate([lens[r][a:b] for r, a, b in pick])
        ax = np.concatenate([auxs[r][a:b] for r, a, b in pick]).astype(np.uint32)
        sl = np.full(len(ln), self.chunk, np.uint32)
        stream = eng.rap_assemble(self.codec, self.gather_buf, off, ln, sl, ax if self.name == "lz4" else None)
        want = torch.cat([self.src[(self.nchunks - half) * self.chunk:], self.shard_of(peer)[:half * self.chunk]])
        got, r = eng.rap_decompress(self.codec, stream, int(want.numel()))
        assert r == want.numel(), f"rank {self.rank}: assembled RAP stream decodes to {r} bytes, expected {want.numel()}"
        assert torch.equal(got[:r], want), f"rank {self.rank}: assembled RAP stream does not decode to the two ranks' bytes"
        return int(stream.numel()), 2 * half

    def verify_slabs(self):
        """The WHOLE sharded stream as a product: every chunk of every rank, in stream order, assembled into the sequence of
        RAP streams the container allows (aoclgpu_rap_assemble_slabs: 1 GiB of source per slab), from the gathered buffer and
        size tables alone; the first and the last slab are decoded and compared with the ranks' bytes."""
        import numpy as np
        torch, eng = self.torch, self.eng
        lens = self.len_all.cpu().numpy().reshape(self.world, self.nchunks)
        auxs = self.aux_all.cpu().numpy().reshape(self.world, self.nchunks)
        off = np.concatenate([np.cumsum(lens[r].astype(np.int64)) - lens[r] + self.seg_off[r] for r in range(self.world)]).astype(np.uint64)
        ln = lens.reshape(-1)
        sl = np.full(len(ln), self.chunk, np.uint32)
        stream, soff, sfirst = eng.rap_assemble_slabs(self.codec, self.gather_buf, off, ln, sl, auxs.reshape(-1).astype(np.uint32) if self.name == "lz4" else None)
        nslabs = len(soff) - 1
        for k in sorted({0, nslabs - 1}):
            c0, c1 = sfirst[k], sfirst[k + 1]
            want = torch.cat([self.shard_of(c // self.nchunks)[(c % self.nchunks) * self.chunk:(c % self.nchunks + 1) * self.chunk] for c in (c0, c1 - 1)])
            got, r = eng.rap_decompress(self.codec, stream[soff[k]:soff[k + 1]], (c1 - c0) * self.chunk)
            assert r == (c1 - c0) * self.chunk, f"slab {k} decodes to {r} bytes"
            assert torch.equal(got[:self.chunk], want[:self.chunk]) and torch.equal(got[r - self.chunk:r], want[self.chunk:]), f"slab {k} differs from the source"
        return nslabs, int(stream.numel())

    def result(self, elapsed):
        """Per-codec record (rank 0 prints it); collective when world > 1."""
        torch, dist, args = self.torch, self.dist, self.args
        comp_bytes = int(self.clen.sum().item())
        ms_c = sum(self.kt["c"]) / len(self.kt["c"])
        ms_d = sum(self.kt["d"]) / len(self.kt["d"])
        comp_total = float(comp_bytes)
        if self.world > 1:
            agg = torch.tensor([ms_c, ms_d, float(comp_bytes)], dtype=torch.float64, device=self.eng.device)
            mx = agg.clone()
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            dist.all_reduce(agg, op=dist.ReduceOp.SUM)
            ms_c, ms_d = float(mx[0]), float(mx[1])
            comp_total = float(agg[2])
        world, total, chunk = self.world, self.total, self.chunk
        unc_total = float(total) * world
        ms_step = elapsed / args.steps * 1e3
        # dominant kernel = the slower of the two launches; algorithmic bytes per launch = N + C
        dom = "compress" if ms_c >= ms_d else "decompress"
        alg_bytes = float(total) + float(comp_bytes)
        dom_ms = ms_c if dom == "compress" else ms_d
        ach = alg_bytes / (dom_ms * 1e-3) / 1e9
        fam = "zstd" if self.name == "zstd3" else self.name
        kernel = ("snappy_compress_mix_kernel" if (self.name, dom) == ("snappy", "compress") and self.nchunks > 16 * 256
                  else f"{fam}_{dom}_kernel" + ("<double-fast>" if (self.name, dom) == ("zstd3", "compress") else ""))
        launches = None
        if (self.name, dom) == ("zstd3", "compress") and self.nchunks > 16 * 256:
            # large level-3 batches: finder kernel + entropy kernel per slab of <= 32 Ki chunks (csrc/zstd_kernels.hip df_split_launch);
            # avg_launch_ms below is the whole compress call = the sum of its launches (profiles/r03_zstd3_*_kernel_trace_and_pmc.txt)
            nslab = self.df_slabs                 # what the launcher used in the timed steps (free memory decides the slab size)
            if nslab:
                kernel = "zstd_df_find_kernel"
                launches = {"per_compress_call": {"zstd_df_find_kernel": nslab, "zstd_df_entropy_kernel": nslab},
                            "note": "avg_launch_ms = one compress call = all of these launches back to back"}
            else:                                 # the launcher fell back to the fused kernel (scratch not available): say so
                launches = {"per_compress_call": {"zstd_compress_kernel<double-fast>": 1},
                            "note": "df_split_launch did not run (workspace not available); the fused double-fast kernel did"}
        out = {
            "value": round(unc_total / GIB / (elapsed / args.steps), 3), "unit": "GiB/s", "ms_per_step": round(ms_step, 3),
            "config": {
                "workload": f"{fam} level-{ZSTD_LEVEL.get(self.name, 1)}, {chunk // 1024} KiB independent chunks, {self.corpus_name} "
                            f"({int(args.base_mib)} MiB base) concatenated to {args.gib:g} GiB per GPU, HBM-resident",
                "codec": self.name, "chunk_bytes": chunk, "chunks_per_gpu": self.nchunks, "gib_per_gpu": args.gib,
                "step": "compress" + (f" -> compact -> exchange of compressed segments between all ranks ({'aoclgpu_exchange_segments (C, RCCL)' if args.exchange == 'c' else args.exchange}, "
                                      "overlapped with) " if world > 1 else " -> ") + "decompress",
                "parallelism": f"chunks sharded over {world} GPU(s), one wavefront per chunk",
            },
            "compress_GiBps": round(unc_total / GIB / (ms_c * 1e-3), 3),
            "decompress_GiBps": round(unc_total / GIB / (ms_d * 1e-3), 3),
            "ratio": round(unc_total / comp_total, 4),
            "roofline": {
                "bound": "hbm", "kernel": kernel, "achieved": round(ach, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBPS, 5), "traffic": None, "algorithmic_bytes_per_launch": alg_bytes,
                "avg_launch_ms": round(dom_ms, 4), "other_kernel": f"{fam}_{'decompress' if dom == 'compress' else 'compress'}_kernel",
                "other_kernel_ms": round(ms_d if dom == "compress" else ms_c, 4),
                "other_kernel_frac": round(alg_bytes / ((ms_d if dom == "compress" else ms_c) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
            },
        }
        if launches:
            out["roofline"]["launches"] = launches
        if world > 1:
            out.update(self.exchange_report(ms_step, ms_c, ms_d))
        # HBM-side traffic of the dominant kernel: not measurable from inside this process; taken from the committed
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command (profiles/traffic.json), raw counter
        # bytes per launch, only when this run has the profiled shape.
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                t = json.load(f).get(self.name)
            out["roofline"]["kernel_source_sha16"] = kernel_source_sha16(self.name)
            if t and t["kernel"] == kernel and t["gib_per_gpu"] == args.gib and t["chunk_bytes"] == chunk:
                if t.get("kernel_source_sha16") == out["roofline"]["kernel_source_sha16"]:
                    out["roofline"]["traffic"] = (t["fetch_kb"] + t["write_kb"]) * 1000.0
                    out["roofline"]["traffic_source"] = t["source"]
                    if t.get("launches"):
                        out["roofline"]["traffic_launches"] = t["launches"]
                else:
                    out["roofline"]["traffic_stale"] = (f"profiles/traffic.json holds {(t['fetch_kb'] + t['write_kb']) * 1000.0:.0f} bytes per launch measured on kernel sources "
                                                        f"{t.get('kernel_source_sha16', '(not recorded)')}; this run's are {out['roofline']['kernel_source_sha16']}")
        except (OSError, ValueError, KeyError):
            pass
        return out


def main():
    args = parse()
    if args.cpu_worker:
        return cpu_worker(json.loads(args.cpu_worker))
    args.exchange_requested = args.exchange
    args.exchange_fallbacks = []
    if args.gpus > 1 and "RANK" not in os.environ:
        # another launcher's rank variable without torch's RANK: every one of ITS processes would start N ranks of its own
        foreign = [k for k in ("OMPI_COMM_WORLD_RANK", "PMI_RANK", "SLURM_PROCID", "LOCAL_RANK") if k in os.environ]
        if foreign and os.environ.get("AOCLGPU_BENCH_LAUNCH_ANYWAY") != "1":
            print(json.dumps({"error": "bench.py --gpus N started under a launcher that does not set RANK / WORLD_SIZE / MASTER_* "
                              f"({', '.join(foreign)} present): run it once without a launcher, or through python -m torch.distributed.run"}), flush=True)
            sys.exit(2)
        sys.exit(self_launch(args))
    import numpy as np
    import torch
    import aocl_compression_amd as a

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # AOCLGPU_BENCH_BACKEND=gloo + AOCLGPU_BENCH_ONE_GPU=1 run the N>1 code path on a 1-GPU box (tests/test_dist_gpu.py)
        backend = os.environ.get("AOCLGPU_BENCH_BACKEND", "nccl")
        if os.environ.get("AOCLGPU_BENCH_ONE_GPU") == "1":
            local = 0
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    assert args.gpus == world, f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks"

    eng = a.Engine(local)
    names = [args.codec] if args.codec != "all" else (["lz4", "snappy", "zstd", "zstd3"] if world == 1 else ["lz4", "zstd"])
    nbase = int(args.base_mib * (1 << 20))
    corpora = {}

    def corpus_for(name):
        key = "jsonlog" if name == "snappy" else "mix"
        if key not in corpora:
            corpora[key] = (a.corpus.make("jsonlog", nbase, seed=0xA0C3), "synthetic JSON/log records") if key == "jsonlog" \
                else (a.corpus.silesia_mix(nbase, seed=0xA0C1), "synthetic Silesia-shaped corpus")
        return corpora[key]

    results = {}
    for name in names:
        base_np, corpus_name = corpus_for(name)
        run = Run(eng, dist, args, name, base_np, corpus_name, rank, world)
        elapsed = run.timed()
        checks = {}
        if not args.no_verify:
            run.verify_local()
            checks["round_trip"] = "every chunk decoded back to its source bytes"
            if world > 1:
                peer = run.verify_remote()
                checks["exchange"] = (f"each rank decoded the segment it received from rank+1 and compared it with that "
                                      f"rank's shard (rank 0 checked rank {peer})")
                nbytes, parts = run.verify_assemble()
                checks["rap_assemble"] = (f"one RAP stream of {parts} partitions across the rank boundary ({nbytes} bytes) assembled "
                                          "from the gathered segments, decoded by aoclgpu_rap_unpack, equal to both ranks' bytes")
                if rank == 0:
                    try:
                        nsl, sbytes = run.verify_slabs()
                        checks["rap_slabs"] = (f"the whole sharded stream ({world} x {run.nchunks} chunks) assembled into {nsl} RAP slab streams "
                                               f"({sbytes} bytes) by aoclgpu_rap_assemble_slabs; first and last slab decoded and compared")
                    except Exception as e:  # noqa: BLE001  (memory on small configurations)
                        checks["rap_slabs"] = {"error": repr(e)}
        res = run.result(elapsed)
        res["checks"] = checks
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            try:
                cpu, cpu_len = cpu_baseline(name, base_np, run.total, run.chunk, run.slot, args.cpu_sample_gib, args.cpu_reps)
                # ratio parity on the sample: GPU sizes == reference sizes, chunk by chunk
                gl = run.clen[:len(cpu_len)].cpu().numpy()
                cpu["sizes_identical_to_gpu"] = bool(np.array_equal(gl, cpu_len))
                res["cpu_baseline"] = cpu
            except Exception as e:  # the baseline is a report, never a reason to lose the bench line
                res["cpu_baseline"] = {"error": repr(e)}
        results[name] = res
        del run
        torch.cuda.empty_cache()
        if rank == 0 and world == 1 and not args.no_by_kind and (name == "lz4" or args.by_kind_all):
            try:
                res["by_kind"] = by_kind(eng, name)
            except Exception as e:
                res["by_kind"] = {"error": repr(e)}

    if rank == 0:
        head = results[names[0]]
        out = {
            "metric": "GiB/s compress + decompress (Silesia mix), ratio parity",
            "value": head["value"], "unit": "GiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic", "config": head["config"],
            "compress_GiBps": head["compress_GiBps"], "decompress_GiBps": head["decompress_GiBps"], "ratio": head["ratio"],
            "roofline": head["roofline"],
        }
        if "cpu_baseline" in head:
            out["cpu_baseline"] = head["cpu_baseline"]
        if world > 1:                                              # the headline codec's exchange record, at the top level too
            for k in ("exchange", "exchange_requested", "exchange_fell_back", "exchange_fallback_reason", "rccl_nranks",
                      "torch_distributed_backend", "exchange_ms", "exchange_GBps", "xgmi_inbound_peak_GBps_per_gpu", "exchange_exposed_ms"):
                out[k] = head.get(k)
            out["launched_by"] = "bench.py itself (child torch.distributed.run)" if os.environ.get("AOCLGPU_BENCH_SELF_LAUNCHED") == "1" else "external launcher"
        out["codecs"] = results
        if world == 1:
            try:
                probes = memory_probes(eng)
                out["hbm_copy_GBps"] = probes.get("hbm_copy_GBps")
                out["hbm_copy_variant"] = probes.get("hbm_copy_variant")
                out["fetch_calibration"] = probes.get("fetch_calibration")
                for r in [out["roofline"]] + [v["roofline"] for v in results.values()]:
                    r["measured_copy_GBps"] = probes.get("hbm_copy_GBps")
                    if probes.get("hbm_copy_GBps"):
                        r["frac_of_measured_copy"] = round(r["achieved"] / probes["hbm_copy_GBps"], 5)
            except Exception as e:
                out["hbm_copy_GBps"] = None
                out["fetch_calibration"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline and args.codec in ("all", "lz4"):
            try:
                out["config1_latency_us"] = config1_latency(eng)
            except Exception as e:
                out["config1_latency_us"] = {"error": repr(e)}
            try:
                out["two_batches_in_flight"] = two_batches_in_flight(eng, args, corpus_for("lz4")[0])
            except Exception as e:
                out["two_batches_in_flight"] = {"error": repr(e)}
            torch.cuda.empty_cache()
            try:
                out["whole_input_frames"] = whole_input_frames(eng)
            except Exception as e:
                out["whole_input_frames"] = {"error": repr(e)}
            try:
                out["host_path"] = host_path(eng)
                cb = out.get("cpu_baseline") or {}
                if isinstance(cb.get("all_threads"), dict):
                    out["host_path"]["reference_all_threads"] = cb["all_threads"]
            except Exception as e:
                out["host_path"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    for f in _BASE_FILES.values():
        try:
            os.unlink(f)
        except OSError:
            pass
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
