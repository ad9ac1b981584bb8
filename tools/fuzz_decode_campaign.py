#!/usr/bin/env python3
"""Random-seed DECODER campaign on the GPU box (TEST INFRASTRUCTURE: loads oracle/): valid streams of all three codecs
(several data kinds, sizes and - for Zstd - levels of the reference-pinned oracle) are mutated (byte flips, truncation,
appended bytes, wrong output capacity) and decoded as explicit batches of 256 streams per launch.  For every stream the GPU
must agree with the oracle: same accept / reject decision and, when accepted, the same bytes.
usage: tools/fuzz_decode_campaign.py <seconds> [seed]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import aocl_compression_amd as a
import oracle_lib as ol

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
eng = a.engine(0); orc = ol.Oracle()
CODECS = [("lz4", a.LZ4, ol.LZ4), ("snappy", a.SNAPPY, ol.SNAPPY), ("zstd", a.ZSTD, ol.ZSTD)]
KINDS = list(a.corpus.KINDS)
t_end = time.time() + budget
it = n_streams = n_accept = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed0 + it); it += 1
    name, gc, oc = CODECS[it % 3]
    streams, caps = [], []
    for k in range(256):
        n = int(rng.choice([0, 1, 20, 300, 5000, 40000, 70000, 131072]))
        raw = a.corpus.make(KINDS[int(rng.integers(0, len(KINDS)))], n, seed=int(rng.integers(0, 1 << 30))).tobytes() if n else b""
        comp = bytearray(orc.compress(oc, raw, level=int(rng.choice([1, 2]))) if oc == ol.ZSTD else orc.compress(oc, raw))
        cap = n
        m = int(rng.integers(0, 8))
        if m <= 3 and comp:                                   # 1-3 byte flips
            for _ in range(m if m else 1):
                comp[int(rng.integers(0, len(comp)))] = int(rng.integers(0, 256))
        elif m == 4 and len(comp) > 2:
            comp = comp[:int(rng.integers(1, len(comp)))]     # truncation
        elif m == 5:
            comp += bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))   # trailing bytes
        elif m == 6:
            cap = max(0, n + int(rng.integers(-5, 6)))        # wrong capacity
        # m == 7: untouched
        streams.append(bytes(comp)); caps.append(cap)
    K = len(streams)
    slen = np.array([len(s) for s in streams], np.uint32)
    soff = np.concatenate([[0], np.cumsum((slen.astype(np.int64) + 15) & ~15)[:-1]]).astype(np.uint64)
    dcap = np.array(caps, np.uint32)
    doff = np.concatenate([[0], np.cumsum((dcap.astype(np.int64) + 63) & ~63)[:-1]]).astype(np.uint64)
    blob = np.zeros(int(soff[-1]) + ((int(slen[-1]) + 15) & ~15) + 64, np.uint8)
    for s, o in zip(streams, soff):
        blob[int(o):int(o) + len(s)] = np.frombuffer(s, np.uint8)
    t = torch; dev = eng.device
    d_src = t.from_numpy(blob).to(dev)
    d_dst = t.zeros(int(doff[-1]) + int(dcap[-1]) + 128, dtype=t.uint8, device=dev)
    d_so = t.from_numpy(soff.view(np.int64)).to(dev); d_sl = t.from_numpy(slen.view(np.int32)).to(dev)
    d_do = t.from_numpy(doff.view(np.int64)).to(dev); d_dc = t.from_numpy(dcap.view(np.int32)).to(dev)
    out_len = t.zeros(K, dtype=t.int32, device=dev)
    b = a.Batch(src=d_src.data_ptr(), dst=d_dst.data_ptr(), src_off=d_so.data_ptr(), src_len=d_sl.data_ptr(), dst_off=d_do.data_ptr(),
                dst_cap=d_dc.data_ptr(), out_len=out_len.data_ptr(), nchunks=K)
    eng._check(eng._fn(gc, "decompress")(C.byref(b), eng._stream()), "decompress")
    res = out_len.cpu().tolist(); host = d_dst.cpu().numpy()
    for k in range(K):
        want = orc.decompress(oc, streams[k], caps[k])
        tag = f"seed {seed0 + it - 1} {name} stream {k} len {len(streams[k])} cap {caps[k]}"
        if want is None:
            assert res[k] < 0, tag + f": GPU accepted ({res[k]}) what the oracle rejects"
        else:
            assert res[k] == len(want), tag + f": GPU {res[k]} oracle {len(want)}"
            assert host[int(doff[k]):int(doff[k]) + res[k]].tobytes() == want, tag
            n_accept += 1
    n_streams += K
print(f"decode fuzz campaign OK: {n_streams} streams ({n_accept} accepted by both, {n_streams - n_accept} rejected by both), seeds {seed0}..{seed0 + it - 1}")
