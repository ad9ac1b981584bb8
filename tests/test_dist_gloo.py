"""world_size-2 gloo test of the N>1 path: chunk ranges tile the stream in rank order, and the ordered gather
of per-rank compacted segments + sizes reproduces the single-process stream (here with the CPU oracle standing
in for the kernels -- the collective logic is what is under test)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import aocl_compression_amd as a
from aocl_compression_amd import sharding


def test_chunk_ranges_tile():
    for n in (0, 1, 7, 64, 1000, 32768):
        for w in (1, 2, 3, 4, 8):
            r = [sharding.chunk_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


def _worker(rank, world, port, q):
    import oracle_lib as ol
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = ol.Oracle()
    chunk = 65536
    data = a.corpus.make("text", 9 * chunk + 1234, seed=5).tobytes()
    nchunks = -(-len(data) // chunk)
    lo, hi = sharding.chunk_range(nchunks, rank, world)
    parts = [orc.compress(ol.LZ4, data[i * chunk:(i + 1) * chunk]) for i in range(lo, hi)]
    packed = torch.from_numpy(np.frombuffer(b"".join(parts), np.uint8).copy())
    lens = torch.tensor([len(p) for p in parts], dtype=torch.int32)
    segs, all_lens = sharding.gather_compressed(dist, packed, lens, seg_cap=nchunks * (chunk + 1024))
    stream = b"".join(bytes(s.numpy()) for s in segs)
    sizes = [int(x) for l in all_lens for x in l]
    # decode the gathered stream in order and compare with the original
    out, pos = [], 0
    for i, sz in enumerate(sizes):
        out.append(orc.decompress(ol.LZ4, stream[pos:pos + sz], min(chunk, len(data) - i * chunk)))
        pos += sz
    q.put((rank, b"".join(out) == data and len(sizes) == nchunks))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]
