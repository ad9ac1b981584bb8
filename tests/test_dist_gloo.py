"""world_size-2 gloo tests of the N>1 path on CPU: chunk ranges tile the stream in rank order; the exact-size exchange of
per-rank compacted segments (aocl_compression_amd.sharding.exchange_segments - the function bench.py runs over RCCL) puts
every rank's segment at its place in every rank's gathered buffer; and the padded all-gather variant reproduces the
single-process stream.  The CPU oracle stands in for the kernels here - the collective logic is what is under test;
tests/test_dist_gpu.py drives the same path with the real kernels on the GPU box."""
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


def test_segment_layout():
    offs, total = sharding.segment_layout([5, 16, 0, 33])
    assert offs == [0, 16, 32, 32] and total == 80
    assert sharding.segment_layout([]) == ([], 0)


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
    # (1) padded all-gather
    segs, all_lens = sharding.gather_compressed(dist, packed, lens, seg_cap=nchunks * (chunk + 1024))
    stream = b"".join(bytes(s.numpy()) for s in segs)
    sizes = [int(x) for l in all_lens for x in l]
    # (2) exact-size exchange into the prefix-sum layout
    mine = torch.tensor([packed.numel()], dtype=torch.int64)
    allsz = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allsz, mine)
    seg = [int(x) for x in allsz]
    seg_off, total = sharding.segment_layout(seg)
    gbuf = torch.full((total,), 0xEE, dtype=torch.uint8)
    gbuf[seg_off[rank]:seg_off[rank] + seg[rank]] = packed
    for w in sharding.exchange_segments(dist, rank, world, gbuf, seg, seg_off):
        w.wait()
    stream2 = b"".join(bytes(gbuf[o:o + s].numpy()) for o, s in zip(seg_off, seg))
    gbuf3 = torch.full((total,), 0xEE, dtype=torch.uint8)
    gbuf3[seg_off[rank]:seg_off[rank] + seg[rank]] = packed
    sharding.exchange_segments(dist, rank, world, gbuf3, seg, seg_off, host_staged=True)
    # decode the gathered stream in order and compare with the original
    out, pos = [], 0
    for i, sz in enumerate(sizes):
        out.append(orc.decompress(ol.LZ4, stream[pos:pos + sz], min(chunk, len(data) - i * chunk)))
        pos += sz
    q.put((rank, b"".join(out) == data and len(sizes) == nchunks and stream2 == stream and torch.equal(gbuf3, gbuf)))
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
