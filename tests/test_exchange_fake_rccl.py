"""csrc/exchange.cpp with MORE THAN ONE RANK on a box without GPUs: the library binds RCCL with dlopen("librccl.so.1"), so worker
processes that find tests/fake_rccl.c (built here, test-only) under that name run aoclgpu_comm_unique_id / comm_create /
allgather_words / exchange_segments between real processes over shared memory - rank arithmetic, offsets, the group structure
and the error paths execute with nranks = 2 and 3.  (RCCL itself with N > 1 stays unexecuted until the driver's multi-GPU run.)"""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "_build", "fake_rccl")

WORKER = textwrap.dedent('''
    import ctypes as C, os, sys, time
    rank, world, idfile, libpath = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    lib = C.CDLL(libpath)
    lib.aoclgpu_last_error.restype = C.c_char_p
    def chk(rc, what):
        assert rc == 0, (what, rc, lib.aoclgpu_last_error())
    ident = (C.c_uint8 * 128)()
    if rank == 0:
        chk(lib.aoclgpu_comm_unique_id(ident), "unique_id")
        with open(idfile + ".tmp", "wb") as f: f.write(bytes(ident))
        os.rename(idfile + ".tmp", idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            assert time.time() - t0 < 30
            time.sleep(0.01)
        ident = (C.c_uint8 * 128).from_buffer_copy(open(idfile, "rb").read())
    comm = C.c_void_p()
    chk(lib.aoclgpu_comm_create(ident, world, rank, C.byref(comm)), "comm_create")
    nr = C.c_int(0)
    chk(lib.aoclgpu_comm_count(comm, C.byref(nr)), "comm_count")
    assert nr.value == world, nr.value
    assert lib.aoclgpu_comm_count(None, C.byref(nr)) != 0
    # per-chunk sizes: every rank contributes 5 words
    mine = (C.c_uint32 * 5)(*[1000 * rank + k for k in range(5)])
    allw = (C.c_uint32 * (5 * world))()
    lib.aoclgpu_allgather_words.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
    chk(lib.aoclgpu_allgather_words(comm, mine, allw, 5, world, rank, None), "allgather_words")
    assert list(allw) == [1000 * r + k for r in range(world) for k in range(5)], list(allw)
    # segments of different sizes (one of them empty), 16-byte aligned starts as sharding.segment_layout places them
    sizes = [((r * 7919 + 13) % 5000) if r != 1 or world < 3 else 0 for r in range(world)]
    offs, o = [], 0
    for s in sizes:
        offs.append(o); o = (o + s + 15) & ~15
    buf = (C.c_uint8 * max(o, 16))()
    for i in range(sizes[rank]): buf[offs[rank] + i] = (rank * 31 + i * 7) & 255
    sz = (C.c_uint64 * world)(*sizes); of = (C.c_uint64 * world)(*offs)
    lib.aoclgpu_exchange_segments.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    for rep in range(3):                                        # the same communicator, several rounds
        chk(lib.aoclgpu_exchange_segments(comm, buf, sz, of, world, rank, None), "exchange_segments")
    for r in range(world):
        for i in range(sizes[r]):
            assert buf[offs[r] + i] == (r * 31 + i * 7) & 255, (rank, r, i)
    # error paths: bad rank, no communicator
    assert lib.aoclgpu_exchange_segments(comm, buf, sz, of, world, world, None) != 0
    assert lib.aoclgpu_exchange_segments(None, buf, sz, of, world, rank, None) != 0
    assert lib.aoclgpu_allgather_words(None, mine, allw, 5, world, rank, None) != 0
    chk(lib.aoclgpu_comm_destroy(comm), "comm_destroy")
    print("rank", rank, "ok")
''')


def _build_fake():
    os.makedirs(BUILD, exist_ok=True)
    out = os.path.join(BUILD, "librccl.so.1")
    src = os.path.join(ROOT, "tests", "fake_rccl.c")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.run(["gcc", "-O1", "-shared", "-fPIC", "-o", out, src, "-lrt"], check=True)
    return BUILD


@pytest.mark.parametrize("world", [2, 3])
def test_c_exchange_between_processes_over_a_fake_rccl(world, tmp_path):
    import aocl_compression_amd as a
    libdir = _build_fake()
    lib = a.api.lib_path()
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = libdir + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    idfile = str(tmp_path / "id.bin")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), idfile, lib], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=120)
        except subprocess.TimeoutExpired:
            for q in procs: q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in out, out
