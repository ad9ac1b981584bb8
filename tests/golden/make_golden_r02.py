#!/usr/bin/env python3
"""Generate tests/golden/golden_r02.json from the REAL reference (oracle/_ref/libaocl_ref.so): round-2 additions to the
golden vectors.  Run in the build container only:  python tests/golden/make_golden_r02.py

Per (corpus kind, size, seed): the reference's ZSTD_compress at levels 3 and 4 (double-fast; size + sha256, and the
compression parameters the reference derived for that level and size, so that the oracle can be driven without the
reference) and LZ4_compress_fast at a few accelerations.  Data only -- no reference source."""
import ctypes as C
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import aocl_compression_amd as a  # noqa: E402
import oracle_lib as ol  # noqa: E402

CASES = [(kind, size, 0xA0C2) for kind in a.corpus.KINDS for size in (20000, 131072)]
CASES += [("text", size, 11) for size in (27, 28, 100, 1000, 5000, 16384, 16385, 65536, 100001)]
CASES += [("mix", 131072, 0xA0C1)]
ACCELS = (2, 8, 65, 1000)


class CP(C.Structure):
    _fields_ = [(k, C.c_uint) for k in ("windowLog", "chainLog", "hashLog", "searchLog", "minMatch", "targetLength")] + [("strategy", C.c_int)]


def data_for(kind, size, seed):
    if kind == "mix":
        return a.corpus.silesia_mix(size, seed=seed).tobytes()
    return a.corpus.make(kind, size, seed=seed).tobytes()


def main():
    ref = ol.Ref()
    L = ref.L
    L.ZSTD_getCParams.restype = CP
    L.ZSTD_getCParams.argtypes = [C.c_int, C.c_ulonglong, C.c_size_t]
    L.aocl_setup_zstd_encode.restype = C.c_void_p
    L.aocl_setup_zstd_encode.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t]
    L.aocl_setup_zstd_encode(0, 3, 0, 3, 0)          # the dispatcher state ZSTD_compress itself establishes (AOCL level table)
    L.LZ4_compress_fast.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int]
    out = {"reference": "amd/aocl-compression 4.2.0 (oracle/_ref, Release flags, all parity-changing options OFF)", "vectors": []}
    for kind, size, seed in CASES:
        raw = data_for(kind, size, seed)
        v = {"kind": kind, "size": size, "seed": seed, "input_sha256": hashlib.sha256(raw).hexdigest(), "zstd": {}, "lz4_fast": {}}
        for level in (3, 4):
            cp = L.ZSTD_getCParams(level, size, 0)
            c = ref.compress(ol.ZSTD, raw, level=level)
            v["zstd"][str(level)] = {"size": len(c), "sha256": hashlib.sha256(c).hexdigest(), "strategy": cp.strategy,
                                     "hashLog": cp.hashLog, "chainLog": cp.chainLog, "minMatch": cp.minMatch}
        cap = ol.bound(ol.LZ4, size)
        for acc in ACCELS:
            buf = C.create_string_buffer(cap)
            r = L.LZ4_compress_fast(raw, buf, size, cap, acc)
            v["lz4_fast"][str(acc)] = {"size": r, "sha256": hashlib.sha256(buf.raw[:r]).hexdigest()}
        out["vectors"].append(v)
    with open(os.path.join(HERE, "golden_r02.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(out["vectors"]), "vectors")


if __name__ == "__main__":
    main()
