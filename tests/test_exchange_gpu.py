"""BASELINE config 5 through the C ABI (csrc/exchange.cpp, rap.hip): the RCCL communicator and the segment exchange of the
library itself, the slab-wise RAP assembly of a stream of any length, and aocl_llc_compress(ZSTD, numThreads > 1) at levels
2-4 in both partition geometries.  One GPU is all the box has: the communicator is exercised with one rank in-process and,
when RCCL lets two processes share the device, with two ranks; the data path of a multi-rank run is covered by
tests/test_dist_gpu.py (host-staged) and the assembly by the tests below."""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

import aocl_compression_amd as a
import oracle_lib as ol

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_comm_single_rank(eng):
    lib = eng.lib
    ident = (C.c_uint8 * 128)()
    eng._check(lib.aoclgpu_comm_unique_id(ident), "unique_id")
    assert any(ident)
    comm = C.c_void_p()
    eng._check(lib.aoclgpu_comm_create(ident, 1, 0, C.byref(comm)), "create")
    assert comm.value
    mine = torch.arange(1000, dtype=torch.int32, device=eng.device)
    allv = torch.zeros(1000, dtype=torch.int32, device=eng.device)
    eng._check(lib.aoclgpu_allgather_words(comm, mine.data_ptr(), allv.data_ptr(), 1000, 1, 0, eng._stream()), "allgather")
    buf = torch.arange(4096, dtype=torch.int32, device=eng.device).to(torch.uint8)
    keep = buf.clone()
    sz = (C.c_uint64 * 1)(4096); of = (C.c_uint64 * 1)(0)
    eng._check(lib.aoclgpu_exchange_segments(comm, buf.data_ptr(), sz, of, 1, 0, eng._stream()), "exchange")
    torch.cuda.synchronize()
    assert torch.equal(allv, mine) and torch.equal(buf, keep)
    eng._check(lib.aoclgpu_comm_destroy(comm), "destroy")
    assert lib.aoclgpu_exchange_segments(None, buf.data_ptr(), sz, of, 2, 0, eng._stream()) != 0      # no communicator: an error, never a hang


_RANK = r'''
import ctypes as C, json, os, sys, time
sys.path.insert(0, sys.argv[1])
import torch
import aocl_compression_amd as a
rank, world, idfile = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
eng = a.engine(0); lib = eng.lib
ident = (C.c_uint8 * 128)()
if rank == 0:
    assert lib.aoclgpu_comm_unique_id(ident) == 0
    open(idfile + ".tmp", "wb").write(bytes(ident)); os.rename(idfile + ".tmp", idfile)
else:
    for _ in range(600):
        if os.path.exists(idfile): break
        time.sleep(0.1)
    ident = (C.c_uint8 * 128).from_buffer_copy(open(idfile, "rb").read())
comm = C.c_void_p()
rc = lib.aoclgpu_comm_create(ident, world, rank, C.byref(comm))
if rc != 0:
    print(json.dumps({"skip": lib.aoclgpu_last_error().decode()})); sys.exit(0)
sizes = [1000 + 7 * r for r in range(world)]
offs = [0]
for s in sizes[:-1]: offs.append(offs[-1] + (s + 15) // 16 * 16)
buf = torch.zeros(offs[-1] + sizes[-1], dtype=torch.uint8, device=eng.device)
buf[offs[rank]:offs[rank] + sizes[rank]] = (torch.arange(sizes[rank], device=eng.device) * (rank + 3) % 251).to(torch.uint8)
sz = (C.c_uint64 * world)(*sizes); of = (C.c_uint64 * world)(*offs)
assert lib.aoclgpu_exchange_segments(comm, buf.data_ptr(), sz, of, world, rank, eng._stream()) == 0, lib.aoclgpu_last_error()
mine = torch.full((5,), rank + 1, dtype=torch.int32, device=eng.device)
allv = torch.zeros(5 * world, dtype=torch.int32, device=eng.device)
assert lib.aoclgpu_allgather_words(comm, mine.data_ptr(), allv.data_ptr(), 5, world, rank, eng._stream()) == 0
torch.cuda.synchronize()
ok = True
for r in range(world):
    want = (torch.arange(sizes[r], device=eng.device) * (r + 3) % 251).to(torch.uint8)
    ok = ok and bool(torch.equal(buf[offs[r]:offs[r] + sizes[r]], want)) and bool((allv[5 * r:5 * r + 5] == r + 1).all())
lib.aoclgpu_comm_destroy(comm)
print(json.dumps({"ok": ok}))
'''


def test_two_ranks_share_the_gpu_over_rccl():
    """two processes, both on GPU 0, one RCCL communicator made by the library: exact-size exchange + all-gather.  RCCL
    builds that refuse two ranks on one device make this a skip (the refusal itself comes back as an error, not a hang)."""
    with tempfile.TemporaryDirectory() as d:
        idf = os.path.join(d, "id")
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
        ps = [subprocess.Popen([sys.executable, "-c", _RANK, ROOT, str(r), "2", idf], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
              for r in range(2)]
        outs = []
        for p in ps:
            try:
                o, e = p.communicate(timeout=240)
            except subprocess.TimeoutExpired:
                for q in ps:
                    q.kill()
                pytest.skip("RCCL did not bring up two ranks on one device within 4 minutes")
            outs.append((p.returncode, o, e))
    recs = []
    for rc, o, e in outs:
        lines = [l for l in o.splitlines() if l.startswith("{")]
        if rc != 0 or not lines:
            pytest.skip("two ranks on one device: " + (e[-300:] or o[-300:]))
        recs.append(json.loads(lines[-1]))
    if any("skip" in r for r in recs):
        pytest.skip("RCCL refused two ranks on one device: " + str([r.get("skip") for r in recs]))
    assert all(r.get("ok") for r in recs), recs


@pytest.mark.parametrize("codec", ["lz4", "snappy", "zstd"])
def test_rap_assemble_slabs(eng, codec):
    """40 chunks' worth of stream cut into slabs of at most 8 chunks: every slab is a complete RAP stream that the GPU decoder
    and the reference's threaded build read; together they are the source."""
    gc = {"lz4": a.LZ4, "snappy": a.SNAPPY, "zstd": a.ZSTD}[codec]
    chunk, n = 65536, 40 * 65536 + 12345
    data = a.corpus.silesia_mix(n, seed=31)
    src = torch.from_numpy(data).to(eng.device)
    slots, out_len, slot = eng.compress_chunks(gc, src, chunk)
    k = out_len.numel()
    aux = None
    if codec == "lz4":                                         # the final literal run of every chunk (the aux[] side output)
        auxt = torch.zeros(k, dtype=torch.int32, device=eng.device)
        b = eng.uniform_compress_batch(src, chunk, slots, out_len, slot)
        b.aux = auxt.data_ptr()
        eng._check(eng._fn(gc, "compress")(C.byref(b), eng._stream()), "compress")
        torch.cuda.synchronize()
        aux = auxt.cpu().numpy().astype(np.uint32)
    ln = out_len.cpu().numpy()
    off = (np.arange(k, dtype=np.uint64) * slot)
    sl = np.full(k, chunk, np.uint32); sl[-1] = n - (k - 1) * chunk
    stream, soff, sfirst = eng.rap_assemble_slabs(gc, slots, off, ln, sl, aux, slab_src_bytes=8 * chunk)
    assert len(soff) - 1 == 6 and sfirst == [0, 8, 16, 24, 32, 40, 41]
    got = []
    raw = data.tobytes()
    for i in range(len(soff) - 1):
        part = stream[soff[i]:soff[i + 1]]
        nbytes = int(sl[sfirst[i]:sfirst[i + 1]].sum())
        dst, r = eng.rap_decompress(gc, part, nbytes)
        assert r == nbytes
        got.append(bytes(dst[:r].cpu().numpy()))
        if ol.have_ref_mt():
            mt = ol.RefMT()
            oc = {"lz4": ol.LZ4, "snappy": ol.SNAPPY, "zstd": ol.ZSTD}[codec]
            assert mt.decompress(oc, bytes(part.cpu().numpy()), nbytes) == got[-1]
    assert b"".join(got) == raw


@pytest.mark.parametrize("level", [2, 3, 4])
def test_llc_zstd_threads_levels(eng, level):
    """aocl_llc_compress(ZSTD, numThreads > 1) at levels 2-4: a RAP stream of 128 KiB partitions, every partition the
    reference's bytes for that level; decodes through aocl_llc_decompress and the reference.  (The ragged tail stays above
    16 KiB: below it the level tables make level 4 the "greedy" strategy, which has no GPU parse.)"""
    data = a.corpus.make("text", 131072 * 5 + 20000, seed=level).tobytes()
    cap = len(data) + (len(data) >> 7) + 4096
    r, blob, _ = eng.llc_compress(a.ZSTD, data, cap, level=level, num_threads=4)
    assert r > 0, eng.lib.aoclgpu_last_error()
    rr, back, _ = eng.llc_decompress(a.ZSTD, blob, len(data))
    assert rr == len(data) and back == data
    if ol.have_ref_mt():
        assert ol.RefMT().decompress(ol.ZSTD, blob, len(data)) == data
    if ol.have_ref():                                         # partition 1 alone is what ZSTD_compress(.., level) writes for it
        ref = ol.Ref()
        ref.L.aocl_destroy_zstd_encode(); ref.L.aocl_setup_zstd_encode(0, 0, 0, 0, 0)
        want = ref.compress(ol.ZSTD, data[131072:262144], level=level)
        assert want in blob


@pytest.mark.parametrize("level", [1, 3])
def test_llc_zstd_threads_reference_geometry(eng, level):
    """AOCLGPU_RAP_GEOMETRY=reference: the bytes of the reference's threaded ZSTD_compress_advanced (the call its own wrapper
    makes, api/codec.cpp:388-422) for the same parameters and thread count: 3 partitions of ~1 MiB, windows of 128 KiB (optVar)"""
    if not ol.have_ref_mt():
        pytest.skip("threaded reference build not available")
    import zstd_frame_cases as zfc
    import test_zstd_api as tza
    mt = ol.RefMT()
    T = 3
    data = zfc.make("text", 3 * (1 << 20) + 4321, seed=5)
    eng.lib.aoclgpu_set_rap_geometry(1)
    try:
        cap = len(data) + (len(data) >> 7) + 4096
        r, blob, _ = eng.llc_compress(a.ZSTD, data, cap, level=level, num_threads=T, opt_var=17)
        assert r > 0, eng.lib.aoclgpu_last_error()
        L = tza._bind(eng.lib)
        p = L.ZSTD_getParams(level, len(data), 0).cParams      # what the wrapper asked for (the dispatcher has run by now)
        # api/codec.cpp:402-408: a window clamped to optVar takes the chain table's log with it (+1 for the binary-tree finders)
        chain = 17 + (1 if p.strategy in (6, 7, 8) else 0) if p.windowLog > 17 else p.chainLog
        cp = zfc.CParams(17, chain, p.hashLog, p.searchLog, p.minMatch, p.targetLength, p.strategy)
        mt.set_threads(T)
        want = zfc.ref_frame(mt, data, cp, extra=4096)
        assert blob == want, (level, len(blob), len(want))
    finally:
        eng.lib.aoclgpu_set_rap_geometry(0)
