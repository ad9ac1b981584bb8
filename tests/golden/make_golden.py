#!/usr/bin/env python3
"""Generate tests/golden/golden.json from the REAL reference (oracle/_ref/libaocl_ref.so, built in place from
/root/reference by oracle/build_ref.sh).  Run in the build container only:  python tests/golden/make_golden.py

Each vector = (corpus kind, size, seed) -> sha256 of the input, and for every codec the reference's compressed
size + sha256 (LZ4_compress_default, snappy_compress, ZSTD_compress level 1).  A few tiny inputs are stored
with their full compressed bytes.  The fixtures hold data only -- no reference source."""
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

CASES = []
for kind in a.corpus.KINDS:
    for size in (65536, 131072, 262144):
        CASES.append((kind, size, 0xA0C1))
for size in (1, 12, 13, 14, 15, 63, 64, 255, 256, 1000, 4096, 16384, 16385, 65535, 65546, 65547, 100001):
    CASES.append(("text", size, 7))
    CASES.append(("binary", size, 9))
CASES.append(("mix", 1 << 20, 0xA0C1))

TINY = [b"", b"a", b"abc", b"a" * 13, b"a" * 100, b"abcd" * 8, b"hello world!", bytes(range(64)) * 3]


def data_for(kind, size, seed):
    if kind == "mix":
        return a.corpus.silesia_mix(size, seed=seed).tobytes()
    return a.corpus.make(kind, size, seed=seed).tobytes()


def main():
    ref = ol.Ref()
    out = {"reference": "amd/aocl-compression 4.2.0 (oracle/_ref, Release flags, all parity-changing options OFF)",
           "codecs": ["lz4", "snappy", "zstd"], "vectors": [], "tiny": []}
    for kind, size, seed in CASES:
        raw = data_for(kind, size, seed)
        v = {"kind": kind, "size": size, "seed": seed, "input_sha256": hashlib.sha256(raw).hexdigest(), "out": {}}
        for name, codec in (("lz4", ol.LZ4), ("snappy", ol.SNAPPY), ("zstd", ol.ZSTD)):
            if codec == ol.ZSTD and size > 131072:
                continue        # the GPU engine's Zstd chunk contract is one block (<= 128 KiB)
            c = ref.compress(codec, raw)
            v["out"][name] = {"size": len(c), "sha256": hashlib.sha256(c).hexdigest()}
            if codec == ol.ZSTD:     # level 2: still the "fast" strategy at these sizes (separate key, older readers ignore it)
                c2 = ref.compress(codec, raw, level=2)
                v.setdefault("zstd_level2", {}).update({"size": len(c2), "sha256": hashlib.sha256(c2).hexdigest()})
        out["vectors"].append(v)
    for raw in TINY:
        t = {"input_hex": raw.hex(), "out": {}}
        for name, codec in (("lz4", ol.LZ4), ("snappy", ol.SNAPPY), ("zstd", ol.ZSTD)):
            t["out"][name] = ref.compress(codec, raw).hex()
        out["tiny"].append(t)
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(out["vectors"]), "vectors,", len(out["tiny"]), "tiny")


if __name__ == "__main__":
    main()
