import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import aocl_compression_amd as a, oracle_lib as ol
eng = a.engine(0); orc = ol.Oracle()
raw = a.corpus.make("text", 1000, seed=1000 + 7).tobytes() if len(sys.argv) < 2 else a.corpus.make(sys.argv[1], int(sys.argv[2]), seed=int(sys.argv[2]) + 7).tobytes()
got = eng.compress_host(a.LZ4, raw); want = orc.compress(ol.LZ4, raw)
print(len(got), len(want), got == want)
def parse(b):
    i = 0; out = []; pos = 0
    while i < len(b):
        t = b[i]; i += 1; ll = t >> 4
        if ll == 15:
            while True:
                s = b[i]; i += 1; ll += s
                if s != 255: break
        i += ll; pos += ll
        if i >= len(b): out.append((pos - ll, ll, None, None)); break
        off = b[i] | b[i + 1] << 8; i += 2; ml = t & 15
        if ml == 15:
            while True:
                s = b[i]; i += 1; ml += s
                if s != 255: break
        ml += 4
        out.append((pos - ll, ll, off, ml)); pos += ml
    return out
pg, pw = parse(got), parse(want)
for k, (x, y) in enumerate(zip(pg, pw)):
    if x != y:
        print("first differing sequence", k, "got", x, "want", y, "previous", pw[k - 1] if k else None); break
