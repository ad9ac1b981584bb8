"""Developer sweep: GPU compress/decompress vs the oracle over corpus kinds x chunk sizes (prints mismatching chunks)."""
import sys, numpy as np, torch, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import aocl_compression_amd as a
import oracle_lib as ol
eng = a.Engine()
orc = ol.Oracle()
codecs = {"lz4": (a.LZ4, ol.LZ4), "snappy": (a.SNAPPY, ol.SNAPPY), "zstd": (a.ZSTD, ol.ZSTD)}
names = sys.argv[1:] or list(codecs)
os.makedirs("gpurun_out", exist_ok=True)
nbad = 0
for name in names:
    gc, oc = codecs[name]
    for kind in list(a.corpus.KINDS):
        for chunk in [700, 1000, 2048, 4096, 5000, 8192, 30000, 65536, 131072] + ([1 << 20] if name != "zstd" else []):
            total = (1 << 20) + 333
            data = a.corpus.make(kind, total, seed=chunk)
            src = torch.from_numpy(data).to(eng.device)
            slots, out_len, slot = eng.compress_chunks(gc, src, chunk)
            lens = out_len.cpu().tolist(); host = slots.cpu().numpy()
            raw = data.tobytes()
            parts = [raw[i:i + chunk] for i in range(0, len(raw), chunk)]
            badc = []
            for i, p in enumerate(parts):
                got = host[i * slot:i * slot + lens[i]].tobytes()
                if got != orc.compress(oc, p):
                    badc.append(i)
                    if len(badc) <= 1:
                        open(f"gpurun_out/bad_{name}_{kind}_{chunk}_{i}.bin", "wb").write(got)
            dst, dl = eng.decompress_chunks(gc, slots, out_len, slot, len(data), chunk)
            neq = int((dst != src).sum().item())
            dlbad = [(i, x) for i, x in enumerate(dl.cpu().tolist()) if x != len(parts[i])]
            if badc or neq or dlbad:
                nbad += 1
                print("BAD", name, kind, chunk, "compress", badc[:5], "decomp mismatches", neq, "dl", dlbad[:3])
print("sweep done, bad configs:", nbad)
