#!/usr/bin/env python3
"""Measurement for the RAP row: end-to-end GiB/s of compress -> aoclgpu_rap_pack (one stream) and of
aoclgpu_rap_unpack on 1 GiB of the bench corpus (RAP offsets are 32-bit: <= 2 GiB per stream), next to the reference's
AOCL_ENABLE_THREADS build on the host cores.  Prints one JSON line per codec."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import aocl_compression_amd as a

GIB = 1 << 30
eng = a.engine(0)
base = a.corpus.silesia_mix(64 << 20)
src = torch.from_numpy(base).to(eng.device).repeat(16)          # 1 GiB
N = src.numel()
for name, codec, chunk in (("lz4", a.LZ4, 262144), ("snappy", a.SNAPPY, 65536), ("zstd", a.ZSTD, 131072)):
    stream = eng.rap_compress(codec, src, chunk)                 # warm-up + result
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        stream = eng.rap_compress(codec, src, chunk)
    torch.cuda.synchronize()
    tc = (time.perf_counter() - t0) / 3
    dst, r = eng.rap_decompress(codec, stream, N)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        dst, r = eng.rap_decompress(codec, stream, N)
    torch.cuda.synchronize()
    td = (time.perf_counter() - t0) / 3
    ok = r == N and bool(torch.equal(dst[:N], src))
    out = {"what": "RAP stream end to end (compress + join / split + decompress), wall clock incl. host plan and allocations",
           "codec": name, "chunk_bytes": chunk, "partitions": -(-N // chunk), "gib": N / GIB,
           "compress_and_pack_GiBps": round(N / GIB / tc, 2), "unpack_GiBps": round(N / GIB / td, 2),
           "stream_bytes": int(stream.numel()), "round_trip_ok": ok}
    try:
        import oracle_lib as ol
        mt = ol.RefMT()
        oc = {"lz4": ol.LZ4, "snappy": ol.SNAPPY, "zstd": ol.ZSTD}[name]
        host = bytes(src[: 256 << 20].cpu().numpy())              # 256 MiB sample
        t0 = time.perf_counter(); s = mt.compress(oc, host, len(host) + len(host) // 6 + (1 << 20)); t1 = time.perf_counter()
        d = mt.decompress(oc, s, len(host)); t2 = time.perf_counter()
        out["reference_mt"] = {"sample_mib": 256, "threads": os.cpu_count(), "compress_GiBps": round(len(host) / GIB / (t1 - t0), 2),
                               "decompress_GiBps": round(len(host) / GIB / (t2 - t1), 2), "ok": d == host}
    except Exception as e:
        out["reference_mt"] = {"error": repr(e)}
    print(json.dumps(out), flush=True)
