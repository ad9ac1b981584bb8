"""Re-run one batch of tools/fuzz_frames_campaign.py by seed and show where a GPU frame and the oracle's part (block by block).
usage: python tools/debug/frame_repro.py <seed> <input index>"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import aocl_compression_amd as a
import oracle_lib as ol
import zstd_frame_cases as zfc
src = open(os.path.join(ROOT, "tools", "fuzz_frames_campaign.py")).read()
ns = {"np": np, "zfc": zfc}
exec(src[src.index("def grammar"):src.index("t0 = time.time()")], ns)
seed, idx = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
inputs = [ns["make_input"](rng) for _ in range(int(rng.integers(1, 9)))]
strategy = int(rng.integers(1, 3))
wl, hl, cl, mm = int(rng.integers(10, 24)), int(rng.integers(6, 19)), int(rng.integers(6, 19)), int(rng.integers(4, 8))
tl = int(rng.integers(0, 2)) if strategy == 1 else 0
data = inputs[idx]
n = len(data)
print("n", n, "params", (strategy, wl, hl, cl, mm, tl))
eng = a.engine(0); orc = ol.Oracle()
f = orc.L.orc_zstd_compress_frame_cp; f.restype = C.c_size_t; f.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int] + [C.c_uint] * 5
out = C.create_string_buffer(n + 4096); r = f(data, n, out, n + 4096, strategy, wl, hl, cl, mm, tl); want = out.raw[:r]
t = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(eng.device)
dst, doff, lens = eng.zstd_compress_frames(t, [0, n], wl, hl, mm, target_length=tl, strategy=strategy, chain_log=cl)
got = dst.cpu().numpy()[:lens[0]].tobytes()
print("gpu", len(got), "oracle", len(want), "equal", got == want)


def blocks(fr):
    fhd = fr[4]; pos = 5 + (0 if (fhd >> 5) & 1 else 1) + [1 if (fhd >> 5) & 1 else 0, 2, 4, 8][fhd >> 6]
    res = []
    while True:
        h = fr[pos] | (fr[pos + 1] << 8) | (fr[pos + 2] << 16); last, typ, sz = h & 1, (h >> 1) & 3, h >> 3
        body = 1 if typ == 1 else sz
        res.append((typ, sz, fr[pos + 3:pos + 3 + body])); pos += 3 + body
        if last: break
    return res


bg, bw = blocks(got), blocks(want)
print("blocks", len(bg), len(bw))
for k, (x, y) in enumerate(zip(bg, bw)):
    if x != y:
        print("first differing block", k, "gpu (type,size)", x[:2], "oracle", y[:2])
        d = next((i for i in range(min(len(x[2]), len(y[2]))) if x[2][i] != y[2][i]), None)
        print("  first differing byte in the block body", d, x[2][:8].hex(), y[2][:8].hex())
        # literals section header of both
        for name, b in (("gpu", x[2]), ("oracle", y[2])):
            t0 = b[0] & 3; sf = (b[0] >> 2) & 3
            print("  ", name, "literals type", t0, "size format", sf, b[:5].hex())
        break


def lit_detail(b):
    v = int.from_bytes(b[:5], "little"); regen = (v >> 4) & 0x3FFFF; comp = (v >> 22) & 0x3FFFF
    hb = b[5]; hs = hb + 1 if hb < 128 else ((hb - 127 + 1) // 2) + 1
    js = 5 + hs
    s1, s2, s3 = (int.from_bytes(b[js + 2 * i:js + 2 * i + 2], "little") for i in range(3))
    return dict(regen=regen, comp=comp, table_header_byte=hb, table_bytes=hs, streams=(s1, s2, s3, comp - hs - 6 - s1 - s2 - s3))


for k, (x, y) in enumerate(zip(bg, bw)):
    if x != y:
        print("gpu   ", lit_detail(x[2])); print("oracle", lit_detail(y[2]))
        gx, gy = lit_detail(x[2]), lit_detail(y[2])
        print("table bytes equal:", x[2][5:5 + gx["table_bytes"]] == y[2][5:5 + gy["table_bytes"]])
        break

back = orc.decompress(ol.ZSTD, got, n)
print("gpu frame decodes to the input (oracle decoder):", back == data)
for k, (x, y) in enumerate(zip(bg, bw)):
    if x != y:
        gx, gy = lit_detail(x[2]), lit_detail(y[2])
        sx, sy = x[2][5 + gx["comp"]:], y[2][5 + gy["comp"]:]
        print("sequence sections equal:", sx == sy, len(sx), len(sy), sx[:4].hex(), sy[:4].hex())
        tx, ty = x[2][5:5 + gx["table_bytes"]], y[2][5:5 + gy["table_bytes"]]
        print("table gpu   ", tx.hex()); print("table oracle", ty.hex())
        break
