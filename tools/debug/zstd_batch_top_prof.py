import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
import aocl_compression_amd as a
eng = a.engine(0)
data = a.corpus.silesia_mix(64 << 20)
src = torch.from_numpy(data).to(eng.device).repeat(8)
gb = (C.c_ulonglong * 4)(); eng.lib.aoclgpu_debug_zstd_gprof(gb)
slots, out_len, slot = eng.compress_chunks(a.ZSTD, src, 131072)
torch.cuda.synchronize(); eng.lib.aoclgpu_debug_zstd_gprof(gb)
n = gb[3]
print("fresh chain batches", n, "| loop top + resume decision %.0f | ring test / refill %.0f | windows + map byte loads (to lgkmcnt) %.0f cycles per batch" % (gb[0] / n, gb[1] / n, gb[2] / n))
