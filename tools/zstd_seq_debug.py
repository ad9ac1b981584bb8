#!/usr/bin/env python3
"""Debug (GPU box): match-finder sequences of the kernel vs the oracle's, first difference per chunk."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import aocl_compression_amd as a
import oracle_lib as ol
eng = a.engine(0)
ol.build_oracle()
O = C.CDLL(ol.ORACLE_SO)
O.orc_dbg_zstd_sequences.restype = C.c_size_t
O.orc_dbg_zstd_sequences.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t)]
eng.lib.aoclgpu_zstd_block_sequences.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32]
expf = int(os.environ.get("ZEXP", "0"))
kinds = sys.argv[1:] or ["text", "jsonlog", "binary", "database"]
nbad = 0
for kind in kinds:
    data = a.corpus.make(kind, 1 << 20).tobytes()
    for i in range(0, len(data), 131072):
        d = data[i:i + 131072]; n = len(d)
        want = np.zeros(4 * (n // 3 + 16), dtype=np.uint32); lit = C.c_size_t(0)
        nb = O.orc_dbg_zstd_sequences(d, n, want.ctypes.data, C.byref(lit))
        want = want[:4 * nb].reshape(-1, 4)[:, :3]
        dev = torch.frombuffer(bytearray(d), dtype=torch.uint8).to(eng.device)
        maxs = n // 3 + 16
        out = np.zeros(2 + 3 * maxs, dtype=np.uint32)
        rc = eng.lib.aoclgpu_zstd_block_sequences(dev.data_ptr(), n, out.ctypes.data, maxs, expf)
        assert rc == 0, rc
        got = out[2:2 + 3 * int(out[0])].reshape(-1, 3)
        if got.shape == want.shape and np.array_equal(got, want) and out[1] == lit.value:
            continue
        nbad += 1
        k = 0
        while k < min(len(got), len(want)) and np.array_equal(got[k], want[k]):
            k += 1
        pos = int(want[:k, 0].sum() + want[:k, 2].sum())
        print(f"{kind} chunk {i >> 17}: nbSeq gpu {out[0]} oracle {nb}; first diff at seq {k} (position {pos}): gpu {got[k:k+3].tolist()} oracle {want[k:k+3].tolist()} prev {want[max(k-2,0):k].tolist()}")
print("bad chunks:", nbad)

# ---- per-batch trace of the first chunk of the first kind, kernel (built with -DZSTD_TRACE) vs the CPU lane model ----
if hasattr(eng.lib, "aoclgpu_debug_zstd_trace"):
    import subprocess
    so = os.path.join(ROOT, "tests", "_build", "libzchain_model.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-I", os.path.join(ROOT, "aocl-compression_amd", "csrc"), os.path.join(ROOT, "tests", "zstd_chain_model.cpp"), "-o", so], check=True)
    Mo = C.CDLL(so)
    Mo.zchain_model_sequences.restype = C.c_size_t
    Mo.zchain_model_sequences.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p, C.c_int]
    d = a.corpus.make(kinds[0], 1 << 20).tobytes()[:131072]; n = len(d)
    dev = torch.frombuffer(bytearray(d), dtype=torch.uint8).to(eng.device)
    gt = np.zeros(8 * 4096, dtype=np.uint64); cnt = C.c_uint(0)
    eng.lib.aoclgpu_debug_zstd_trace.argtypes = [C.c_void_p, C.c_void_p]
    Mo.zchain_model_trace.argtypes = [C.c_void_p]
    eng.lib.aoclgpu_debug_zstd_trace(gt.ctypes.data, C.byref(cnt))          # reset
    out = np.zeros(2 + 3 * (n // 3 + 16), dtype=np.uint32)
    eng.lib.aoclgpu_zstd_block_sequences(dev.data_ptr(), n, out.ctypes.data, n // 3 + 16, expf)
    eng.lib.aoclgpu_debug_zstd_trace(gt.ctypes.data, C.byref(cnt))
    mt = np.zeros(8 * 4096, dtype=np.uint64)
    Mo.zchain_model_trace(mt.ctypes.data)
    o2 = np.zeros(4 * (n // 3 + 16), dtype=np.uint32); lit = C.c_size_t(0)
    Mo.zchain_model_sequences(d, n, o2.ctypes.data, C.byref(lit), None, 1)
    mc = Mo.zchain_model_trace_count()
    gt = gt.reshape(-1, 8); mt = mt.reshape(-1, 8)
    print("trace entries: gpu", cnt.value, "model", mc)
    names = ["pos0", "lim", "okmask", "commit", "outcome|S|ev|el|ntail|post", "D1.lo", "D1.hi", "rep1|rep2"]
    for k in range(min(cnt.value, mc, 4096)):
        if not np.array_equal(gt[k], mt[k]):
            print("first differing batch", k)
            for j in range(max(k - 1, 0), k + 1):
                print(" batch", j)
                for f in range(8):
                    print(f"   {names[f]:28s} gpu {int(gt[j][f]):#018x}  model {int(mt[j][f]):#018x}")
            break
    else:
        print("traces agree on the common prefix")
