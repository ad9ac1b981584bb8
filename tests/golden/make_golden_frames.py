#!/usr/bin/env python3
"""Generate tests/golden/golden_frames.json from the REAL reference (oracle/_ref/libaocl_ref.so): whole-input frames.
Run in the build container only:  python tests/golden/make_golden_frames.py

Per input (tests/zstd_frame_cases.py: kind, size, seed): the reference's ZSTD_compress_advanced frame for a few parameter sets
- the rows ZSTD_getCParams reports for levels 1-4 at that size (AOCL level table) and explicit ones with a window below the
input, a window below the block size, targetLength 1 - and the reference's LZ4F_compressFrame with linked blocks for two
block sizes and two levels.  Size + sha256 of every frame; data only - no reference source."""
import ctypes as C
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as ol  # noqa: E402
import lz4f_lib as lf  # noqa: E402
import zstd_frame_cases as zfc  # noqa: E402

INPUTS = [("text", 131073, 1), ("mixed", 300000, 2), ("jsonlog", 524289, 3), ("low", 200000, 4), ("rep", 262145, 5),
          ("zeros", 400000, 6), ("random", 150000, 7), ("mixed", 1500000, 8), ("text", 70000, 9)]
EXPLICIT = [(1, 17, 13, 14, 1, 6, 0), (1, 12, 13, 13, 1, 5, 0), (1, 19, 12, 13, 1, 6, 1), (2, 17, 15, 16, 1, 5, 0), (2, 12, 10, 11, 1, 4, 0)]
# (strategy, windowLog, chainLog, hashLog, searchLog, minMatch, targetLength)


def main():
    ref = ol.Ref()
    R = lf.bind(ref.L)
    out = {"reference": "amd/aocl-compression 4.2.0 (oracle/_ref, Release flags, all parity-changing options OFF)", "vectors": []}
    for kind, size, seed in INPUTS:
        raw = zfc.make(kind, size, seed)
        v = {"kind": kind, "size": size, "seed": seed, "input_sha256": hashlib.sha256(raw).hexdigest(), "zstd": [], "lz4f_linked": []}
        cps = []
        for level in (1, 2, 3, 4):
            cp = zfc.ref_level_cparams(ref, level, size)
            if cp.strategy <= 2:
                cps.append((f"level {level}", cp))
        for (st, wl, cl, hl, sl, mm, tl) in EXPLICIT:
            cps.append(("explicit", zfc.CParams(wl, cl, hl, sl, mm, tl, st)))
        for tag, cp in cps:
            c = zfc.ref_frame(ref, raw, cp)
            v["zstd"].append({"from": tag, "strategy": cp.strategy, "windowLog": cp.windowLog, "chainLog": cp.chainLog, "hashLog": cp.hashLog,
                              "minMatch": cp.minMatch, "targetLength": cp.targetLength, "size": len(c), "sha256": hashlib.sha256(c).hexdigest()})
        for bsid, level in ((4, 0), (5, 0), (4, -3)):
            c = lf.compress(R, raw, lf.prefs(bsid=bsid, level=level))
            assert isinstance(c, bytes)
            v["lz4f_linked"].append({"blockSizeID": bsid, "level": level, "size": len(c), "sha256": hashlib.sha256(c).hexdigest()})
        out["vectors"].append(v)
    with open(os.path.join(HERE, "golden_frames.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(out["vectors"]), "inputs,", sum(len(v["zstd"]) + len(v["lz4f_linked"]) for v in out["vectors"]), "frames")


if __name__ == "__main__":
    main()
