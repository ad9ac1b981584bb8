#!/usr/bin/env python3
"""Generates tests/golden/lz4f_golden.json from the REFERENCE's lz4frame.c (oracle/_ref/libaocl_ref.so, LZ4F_compressFrame).
Inputs are seeded synthetic data (aocl_compression_amd.corpus); small frames are stored as hex, large ones as sha256 + size.
Run in the build container (needs /root/reference to have been compiled by oracle/build_ref.sh)."""
import ctypes as C, hashlib, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import aocl_compression_amd as a
import oracle_lib as ol
import lz4f_lib as lf
R = lf.bind(C.CDLL(ol.REF_SO))
cases = []
for kind, n, seed in [("text", 0, 1), ("text", 5, 1), ("text", 300, 2), ("binary", 2000, 3), ("random", 1000, 4), ("text", 70000, 5), ("jsonlog", 300000, 6), ("database", 1 << 20, 7)]:
    data = a.corpus.make(kind, max(n, 4096), seed=seed).tobytes()[:n]
    for o in [dict(), dict(independent=True), dict(bsid=5, independent=True, content_checksum=True, block_checksum=True, content_size=True), dict(bsid=7, content_checksum=True)]:
        frame = lf.compress(R, data, lf.prefs(**o))
        assert isinstance(frame, bytes)
        e = {"kind": kind, "n": n, "seed": seed, "prefs": o, "frame_bytes": len(frame), "frame_sha256": hashlib.sha256(frame).hexdigest(),
             "data_sha256": hashlib.sha256(data).hexdigest()}
        if len(frame) <= 2200:
            e["frame_hex"] = frame.hex()
        cases.append(e)
json.dump({"source": "LZ4F_compressFrame of amd/aocl-compression 4.2.0 (algos/lz4/lz4frame.c) via oracle/_ref", "cases": cases},
          open(os.path.join(HERE, "lz4f_golden.json"), "w"), indent=1)
print(len(cases), "cases")
