#!/usr/bin/env python3
"""Random-seed campaign for the RAP join (TEST INFRASTRUCTURE: loads oracle/ and the reference builds): random inputs (corpus
kinds, LZ grammar, prescribed parses, incompressible stretches) cut into partitions of random sizes (13 B - 300 KB: all-literal
partitions, literal runs that cross the 15 / 270 byte token boundaries when merged, ragged tails), compressed and joined on the
GPU (aoclgpu_rap_pack), against the Python restatement of the reference's join (tests/rap_model.py, pinned to the threaded
build), the reference's single-threaded decoder after the frame skip, and the GPU's own RAP decoder.
usage: tools/fuzz_rap_campaign.py <seconds> [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import aocl_compression_amd as a
import oracle_lib as ol
import rap_model
import zstd_frame_cases as zfc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
eng = a.engine(0); orc = ol.Oracle(); ref = ol.Ref(); ref_mt = ol.RefMT()
src_code = open(os.path.join(ROOT, "tools", "fuzz_campaign.py")).read()
ns = {"np": np}
exec(src_code[src_code.index("def build(rng, target):"):src_code.index("def want_lz4")], ns)     # build() and seq_shaped()
CODECS = [("lz4", a.LZ4, ol.LZ4), ("snappy", a.SNAPPY, ol.SNAPPY), ("zstd", a.ZSTD, ol.ZSTD)]
t0 = time.time(); it = 0; n_streams = 0; n_parts = 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed0 + it); it += 1
    name, gc, oc = CODECS[it % 3]
    total = int(rng.choice([rng.integers(1, 3000), rng.integers(3000, 200000), rng.integers(200000, 1500000)]))
    k = int(rng.integers(0, 4))
    if k == 0: data = ns["build"](rng, total)
    elif k == 1: data = ns["seq_shaped"](rng, total)
    elif k == 2: data = zfc.make("mixed", total, int(rng.integers(0, 1 << 30)))
    else: data = a.corpus.make(list(a.corpus.KINDS)[int(rng.integers(0, len(a.corpus.KINDS)))], total, seed=int(rng.integers(0, 1 << 30))).tobytes()
    chunk = int(rng.choice([13, 14, 15, 16, 17, 30, 100, 270, 271, 285, 1000, 4096, 65536, 65547, 131072, 262144, 300000]))
    if rng.random() < 0.3: chunk = int(rng.integers(13, 5000))
    if name == "zstd" and chunk > 131072: chunk = 131072
    if -(-total // chunk) > 20000: chunk = -(-total // 20000)
    src = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(eng.device)
    stream_t = eng.rap_compress(gc, src, chunk)
    stream = bytes(stream_t.cpu().numpy())
    tag = f"seed {seed0 + it - 1} {name} total {total} chunk {chunk}"
    assert stream == rap_model.pack(orc, oc, data, chunk), tag + ": GPU join differs from the model of the reference's join"
    assert ref.decompress(oc, stream[ref_mt.skip(stream):], total) == data, tag + ": the reference's decoder disagrees"
    dst, r = eng.rap_decompress(gc, stream_t, total)
    assert r == total and torch.equal(dst[:total], src), tag + ": GPU RAP decode"
    n_streams += 1; n_parts += -(-total // chunk)
print(f"RAP campaign OK: {n_streams} streams, {n_parts} partitions, seeds {seed0}..{seed0 + it - 1}")
