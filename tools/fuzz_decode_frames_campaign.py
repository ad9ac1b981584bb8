#!/usr/bin/env python3
"""Random-seed DECODER campaign on mutated MULTI-BLOCK Zstd frames (TEST INFRASTRUCTURE: loads oracle/): frames of several
blocks written by the oracle's frame compressor (fast and double-fast, small windows: dozens of blocks, treeless literal
sections, RLE and raw blocks, window descriptors) are mutated (byte flips, truncation, trailing bytes, wrong capacity) and
decoded as explicit batches on the GPU.  For every stream the GPU must agree with the oracle's decoder: same accept / reject
decision and, when accepted, the same bytes.  usage: tools/fuzz_decode_frames_campaign.py <seconds> [seed]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import aocl_compression_amd as a
import oracle_lib as ol
import zstd_frame_cases as zfc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
eng = a.engine(0); orc = ol.Oracle()
f = orc.L.orc_zstd_compress_frame_cp; f.restype = C.c_size_t
f.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int] + [C.c_uint] * 5
KINDS = ["text", "mixed", "low", "rep", "zeros", "jsonlog", "random"]
t_end = time.time() + budget
it = n_streams = n_accept = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed0 + it); it += 1
    streams, caps = [], []
    for k in range(64):
        n = int(rng.choice([rng.integers(1, 5000), rng.integers(5000, 200000), rng.integers(131073, 500000)]))
        raw = zfc.make(KINDS[int(rng.integers(0, len(KINDS)))], n, int(rng.integers(0, 1 << 30)))
        strategy = int(rng.integers(1, 3))
        wl = int(rng.choice([10, 11, 12, 14, 17, 19, 22])); hl = int(rng.integers(7, 16)); cl = int(rng.integers(7, 16)); mm = int(rng.integers(4, 8))
        out = C.create_string_buffer(n + (n >> 8) + 64)
        r = f(raw, n, out, n + (n >> 8) + 64, strategy, wl, hl, cl, mm, 0)
        assert r > 0
        comp = bytearray(out.raw[:r]); cap = n
        m = int(rng.integers(0, 8))
        if m <= 3:
            for _ in range(m if m else 1):
                comp[int(rng.integers(0, len(comp)))] = int(rng.integers(0, 256))
        elif m == 4 and len(comp) > 2:
            comp = comp[:int(rng.integers(1, len(comp)))]
        elif m == 5:
            comp += bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))
        elif m == 6:
            cap = max(0, n + int(rng.integers(-5, 6)))
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
    eng._check(eng._fn(a.ZSTD, "decompress")(C.byref(b), eng._stream()), "decompress")
    res = out_len.cpu().tolist(); host = d_dst.cpu().numpy()
    for k in range(K):
        want = orc.decompress(ol.ZSTD, streams[k], caps[k])
        tag = f"seed {seed0 + it - 1} stream {k} len {len(streams[k])} cap {caps[k]}"
        if want is None:
            assert res[k] < 0, tag + f": GPU accepted ({res[k]}) what the oracle rejects"
        else:
            assert res[k] == len(want), tag + f": GPU {res[k]} oracle {len(want)}"
            assert host[int(doff[k]):int(doff[k]) + res[k]].tobytes() == want, tag
            n_accept += 1
    n_streams += K
print(f"multi-block decode fuzz campaign OK: {n_streams} streams ({n_accept} accepted by both, {n_streams - n_accept} rejected by both), seeds {seed0}..{seed0 + it - 1}")
