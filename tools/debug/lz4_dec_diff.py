#!/usr/bin/env python3
"""Debug aid: decode the kbench LZ4 stream on the GPU, report the first chunks whose output differs from the input and the
sequence (literal length, match length, offset, position) that covers the first wrong byte."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import aocl_compression_amd as m

eng = m.engine(0)
if len(sys.argv) > 2:                                   # lz4_dec_diff.py <file> <chunk> [flags]: an input kept by a fuzz campaign
    raw = open(sys.argv[1], "rb").read(); chunk = int(sys.argv[2]); flags = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0
    total = len(raw)
    src = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(eng.device)
else:
    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    chunk = 262144; flags = 0
    total = int(gib * (1 << 30)) // chunk * chunk
    base = torch.from_numpy(m.corpus.silesia_mix(211_938_580, seed=0xA0C1)).to(eng.device)
    src = base.repeat(-(-total // base.numel()))[:total].contiguous()
slots, clen, slot = eng.compress_chunks(m.LZ4, src, chunk, flags=flags)
dst, dl = eng.decompress_chunks(m.LZ4, slots, clen, slot, total, chunk)
torch.cuda.synchronize()
n = -(-total // chunk)
pad = n * chunk - total
sv = torch.cat([src, torch.zeros(pad, dtype=torch.uint8, device=src.device)]) if pad else src
dv = torch.cat([dst, torch.zeros(pad, dtype=torch.uint8, device=src.device)]) if pad else dst
bad = (dv.view(n, chunk) != sv.view(n, chunk)).any(dim=1).nonzero().flatten().cpu().tolist()
print("chunks", n, "bad", len(bad), bad[:10], "dl", dl.cpu().tolist()[:8])
for c in bad[:3]:
    a = sv[c * chunk:(c + 1) * chunk].cpu().numpy(); b = dv[c * chunk:(c + 1) * chunk].cpu().numpy()
    w = int(np.nonzero(a != b)[0][0]); nw = int((a != b).sum())
    blk = slots[c * slot:c * slot + int(clen[c])].cpu().numpy().tobytes()
    i = 0; o = 0; k = 0
    print(f"chunk {c}: first wrong byte {w}, wrong bytes {nw}, dl {int(dl[c])}")
    hist = []
    while i < len(blk):
        tp = i; t = blk[i]; i += 1; lit = t >> 4
        if lit == 15:
            while True:
                x = blk[i]; i += 1; lit += x
                if x != 255: break
        i += lit
        if i >= len(blk): break
        off = blk[i] | (blk[i + 1] << 8); i += 2; ml = t & 15
        if ml == 15:
            while True:
                x = blk[i]; i += 1; ml += x
                if x != 255: break
        ml += 4
        hist.append((k, tp, o, lit, ml, off))
        if o + lit + ml > w:
            for h in hist[-6:]: print("   seq %d tok@%d out@%d lit %d mlen %d off %d" % h)
            break
        o += lit + ml; k += 1
    print("   want", a[max(0, w - 8):w + 24].tobytes())
    print("   got ", b[max(0, w - 8):w + 24].tobytes())
