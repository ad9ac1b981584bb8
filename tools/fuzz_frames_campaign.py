#!/usr/bin/env python3
"""Random-seed parity campaign for whole-input frames on the GPU box (TEST INFRASTRUCTURE: loads oracle/).
Zstd: batches of LZ-grammar / mixed-statistics inputs of 1 B - 3 MB compressed as frames by aoclgpu_zstd_compress_frames with
random parameters (fast and double-fast, windows from 1 KiB to 8 MiB - below the block size, below the input, above it -
hashLog / chainLog 6-18, minMatch 4-7, targetLength 0-131072) against orc_zstd_compress_frame_cp, and decoded back by the GPU
decoder.  LZ4: LZ4F_compressFrame with linked blocks (block sizes 64 KiB - 4 MiB, accelerations) against the oracle's linked
blocks, decoded back through LZ4F_decompress.  usage: tools/fuzz_frames_campaign.py <seconds> [seed]"""
import ctypes as C, os, struct, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import aocl_compression_amd as a
import oracle_lib as ol
import lz4f_lib as lf
import zstd_frame_cases as zfc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
eng = a.engine(0); orc = ol.Oracle(); L = orc.L; lib = eng.lib; G = lf.bind(lib)
L.orc_zstd_compress_frame_cp.restype = C.c_size_t
L.orc_zstd_compress_frame_cp.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int] + [C.c_uint] * 5
L.orc_lz4_compress_linked_blocks.restype = C.c_int
L.orc_lz4_compress_linked_blocks.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32)]


def grammar(rng, target):
    """LZ-grammar bytes: literal runs over a small alphabet, copies at all distances (also beyond 128 KiB), byte runs"""
    out = bytearray()
    alphabet = int(rng.choice([2, 4, 16, 256]))
    while len(out) < target:
        k = rng.integers(0, 5)
        if k == 0:
            out += bytes(rng.integers(0, alphabet, int(rng.integers(1, 200)), dtype=np.uint8))
        elif k == 1 and out:
            d = 1 + int(rng.integers(0, min(len(out), 1 << 21))); n = int(rng.integers(3, 2000))
            if d >= n: out += out[len(out) - d:len(out) - d + n]
            else:
                seg = bytes(out[-d:]); out += (seg * (n // d + 1))[:n]
        elif k == 2:
            out += bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 200000 if rng.random() < 0.02 else 600))
        elif k == 3 and len(out) >= 9:
            p = int(rng.integers(1, 10)); n = int(rng.integers(4, 300))
            seg = bytes(out[-p:]); out += (seg * (n // p + 1))[:n]
        elif k == 4:
            out += bytes(rng.integers(0, 256, int(rng.integers(1, 5000)), dtype=np.uint8))
    return bytes(out[:target])


def make_input(rng):
    n = int(rng.choice([rng.integers(1, 2000), rng.integers(1000, 200000), rng.integers(131073, 600000), rng.integers(131073, 3000000)]))
    if rng.random() < 0.5:
        return grammar(rng, n)
    return zfc.make("mixed", n, int(rng.integers(0, 1 << 30)))


t0 = time.time(); it = 0; frames = 0; nbytes = 0; lz4 = 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed0 + it); it += 1
    inputs = [make_input(rng) for _ in range(int(rng.integers(1, 9)))]
    strategy = int(rng.integers(1, 3))
    wl, hl, cl, mm = int(rng.integers(10, 24)), int(rng.integers(6, 19)), int(rng.integers(6, 19)), int(rng.integers(4, 8))
    tl = int(rng.choice([0, 1, 1, 2, 3, 4, 7, 8, 31, 127, 128, 129, 1000, 131072])) if strategy == 1 else 0   # step = tl + 1 above 1
    off = np.zeros(len(inputs) + 1, dtype=np.uint64); off[1:] = np.cumsum([len(x) for x in inputs])
    src = torch.frombuffer(bytearray(b"".join(inputs)), dtype=torch.uint8).to(eng.device)
    flags = int(rng.choice([1, 2])) if strategy == 1 and hl <= 14 else 0         # both finders of the fast strategy
    dst, doff, lens = eng.zstd_compress_frames(src, off, wl, hl, mm, target_length=tl, strategy=strategy, chain_log=cl, flags=flags)
    host = dst.cpu().numpy()
    for i, data in enumerate(inputs):
        n = len(data); cap = n + (n >> 8) + 64
        w = C.create_string_buffer(cap)
        r = L.orc_zstd_compress_frame_cp(data, n, w, cap, strategy, wl, hl, cl, mm, tl)
        got = host[doff[i]:doff[i] + lens[i]].tobytes()
        if got != w.raw[:r]:
            print("ZSTD FRAME MISMATCH seed", seed0 + it - 1, "input", i, "n", n, (strategy, wl, hl, cl, mm, tl), "flags", flags, len(got), r); sys.exit(1)
        back = C.create_string_buffer(max(n, 1))
        rr = lib.aoclgpu_zstd_decompress_host_frames(got, len(got), back, n)
        if rr != n or back.raw[:n] != data:
            print("ZSTD FRAME DECODE MISMATCH seed", seed0 + it - 1, "input", i, "n", n, rr); sys.exit(1)
        frames += 1; nbytes += n
    if it % 3 == 0:                                            # LZ4 frame with linked blocks
        data = inputs[int(np.argmax([len(x) for x in inputs]))]
        if len(data) > 65536:
            bsid = int(rng.choice([4, 4, 5, 6, 7])); level = int(rng.choice([0, 0, -1, -9, -200])); bs = 1 << (8 + 2 * bsid)
            opts = dict(content_checksum=bool(rng.integers(0, 2)), block_checksum=bool(rng.integers(0, 2)))
            p = lf.prefs(bsid=bsid, level=level, **opts)
            got = lf.compress(G, data, p)
            n = len(data)
            out = C.create_string_buffer(n + 16); bl = (C.c_uint32 * (n // bs + 2))()
            nb = L.orc_lz4_compress_linked_blocks(data, n, bs, -level + 1 if level < 0 else 1, out, n + 16, bl)
            pos, at, ok = 7, 0, isinstance(got, bytes)
            for k in range(nb if ok else 0):
                (h,) = struct.unpack_from("<I", got, pos); sz = h & 0x7FFFFFFF
                if h != bl[k] or got[pos + 4:pos + 4 + sz] != out.raw[at:at + sz]: ok = False; break
                pos += 4 + sz + (4 if opts["block_checksum"] else 0); at += sz
            if not ok or (n > bs and (got[4] >> 5) & 1):
                print("LZ4F LINKED MISMATCH seed", seed0 + it - 1, "n", n, bsid, level, opts); sys.exit(1)
            if lf.decompress(G, got, n) != (data, len(got)):
                print("LZ4F LINKED DECODE MISMATCH seed", seed0 + it - 1, "n", n, bsid, level); sys.exit(1)
            lz4 += 1
print(f"OK seed0={seed0} batches={it} zstd_frames={frames} bytes={nbytes} lz4f_linked_frames={lz4} in {time.time() - t0:.0f}s")
