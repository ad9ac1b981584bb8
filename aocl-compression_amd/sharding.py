"""Chunk sharding across ranks (the GPU counterpart of the reference's RAP partition split,
threads/threads.c:57-80,121-153): contiguous chunk ranges, ordered gather of compressed segments.

The exchange itself is the C library's (csrc/exchange.cpp: aoclgpu_comm_create / aoclgpu_exchange_segments /
aoclgpu_allgather_words over RCCL, bound at run time); `CExchange` below is its thin binding.  The torch.distributed
functions that follow are the fallbacks bench.py keeps (a grouped isend / irecv, a padded all-gather) and the host-staged
gloo transport of the 1-GPU / CPU validation runs."""
from __future__ import annotations

import ctypes as C


class _StreamWait:
    """what `exchange` hands back: wait() makes the caller's current stream wait for the exchange stream"""

    def __init__(self, torch, side):
        self.torch, self.side = torch, side

    def wait(self):
        self.torch.cuda.current_stream().wait_stream(self.side)


class CExchange:
    """The C exchange (include/aocl_gpu_codec.h, config 5): an RCCL communicator created through the library on this rank's
    device - the 128-byte id travels from rank 0 over whatever torch.distributed backend is up - and the two data-path calls,
    enqueued on a side stream so that they overlap with the rank's own decompression."""

    def __init__(self, eng, dist, rank: int, world: int):
        import torch
        self.torch, self.eng, self.rank, self.world = torch, eng, rank, world
        lib = eng.lib
        # every step that can fail on one rank only is followed by an agreement step, so that all ranks raise together and
        # nobody is left waiting in a collective (the caller then falls back to torch.distributed on every rank)
        ident = (C.c_uint8 * 128)()
        rc = lib.aoclgpu_comm_unique_id(ident) if rank == 0 else 0
        obj = [(int(rc), bytes(ident))]
        dist.broadcast_object_list(obj, src=0)
        rc, raw = obj[0]
        if rc != 0:
            raise RuntimeError("aoclgpu_comm_unique_id failed on rank 0 (is librccl.so.1 loadable?)")
        ident = (C.c_uint8 * 128).from_buffer_copy(raw)
        comm = C.c_void_p()
        rc = int(lib.aoclgpu_comm_create(ident, world, rank, C.byref(comm)))
        rcs = [None] * world
        dist.all_gather_object(rcs, rc)
        if any(rcs):
            if rc == 0 and comm:
                lib.aoclgpu_comm_destroy(comm)
            raise RuntimeError(f"aoclgpu_comm_create failed on ranks {[r for r, v in enumerate(rcs) if v]}")
        self.comm = comm
        self.side = torch.cuda.Stream(device=eng.device)

    def exchange(self, gather_buf, seg_sizes, seg_off):
        sz = (C.c_uint64 * self.world)(*[int(x) for x in seg_sizes])
        of = (C.c_uint64 * self.world)(*[int(x) for x in seg_off])
        self.side.wait_stream(self.torch.cuda.current_stream())          # the compaction into gather_buf comes first
        self.eng._check(self.eng.lib.aoclgpu_exchange_segments(self.comm, gather_buf.data_ptr(), sz, of, self.world, self.rank,
                                                               self.side.cuda_stream), "exchange_segments")
        return [_StreamWait(self.torch, self.side)]

    def allgather_words(self, out, mine):
        """out[r * n ..] = rank r's `mine` (int32 / uint32 device tensors).  Enqueued on the SAME side stream as the segment
        exchange: one communicator never has operations in flight on two streams."""
        cur = self.torch.cuda.current_stream()
        self.side.wait_stream(cur)
        self.eng._check(self.eng.lib.aoclgpu_allgather_words(self.comm, mine.data_ptr(), out.data_ptr(), int(mine.numel()), self.world, self.rank,
                                                             self.side.cuda_stream), "allgather_words")
        cur.wait_stream(self.side)

    def nranks(self):
        """the number of ranks RCCL holds for this communicator (ncclCommCount), or None when the library cannot say"""
        n = C.c_int(0)
        return int(n.value) if self.eng.lib.aoclgpu_comm_count(self.comm, C.byref(n)) == 0 else None

    def close(self):
        if self.comm:
            self.eng.lib.aoclgpu_comm_destroy(self.comm)
            self.comm = None


def chunk_range(nchunks: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous range [lo, hi) of chunks owned by `rank` (ranges tile [0, nchunks) in rank order)."""
    return (rank * nchunks) // world, ((rank + 1) * nchunks) // world


def gather_compressed(dist, packed, lens, seg_cap: int):
    """All-gather of per-rank compacted segments (padded to seg_cap) and their per-chunk sizes.
    Returns (list of per-rank segment tensors trimmed to their true size, list of per-rank length tensors).
    Works with any torch.distributed backend (nccl = RCCL on ROCm, gloo in the CPU tests)."""
    import torch
    world = dist.get_world_size()
    n = torch.tensor([int(lens.numel()), int(lens.clamp(min=0).sum().item())], dtype=torch.int64, device=lens.device)
    meta = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(meta, n)
    maxn = max(int(m[0]) for m in meta)
    lens_pad = torch.zeros(maxn, dtype=lens.dtype, device=lens.device)
    lens_pad[:lens.numel()] = lens
    all_lens = [torch.zeros_like(lens_pad) for _ in range(world)]
    dist.all_gather(all_lens, lens_pad)
    seg = torch.zeros(seg_cap, dtype=torch.uint8, device=packed.device)
    seg[:packed.numel()] = packed
    segs = [torch.zeros_like(seg) for _ in range(world)]
    dist.all_gather(segs, seg)
    return ([s[:int(m[1])] for s, m in zip(segs, meta)], [l[:int(m[0])] for l, m in zip(all_lens, meta)])


def segment_layout(seg_sizes, align: int = 16):
    """Where rank r's compacted segment lives in the gathered buffer: the exclusive prefix sum of the segment sizes with
    `align`-byte aligned starts.  Returns (offsets, total bytes)."""
    offs, o = [], 0
    for s in seg_sizes:
        offs.append(o)
        o += (int(s) + align - 1) // align * align
    return offs, o


def exchange_segments(dist, rank: int, world: int, gather_buf, seg_sizes, seg_off, host_staged: bool = False):
    """Exact-size exchange of the compressed segments between all ranks (RCCL has no all-gatherv): this rank's segment
    already lies at gather_buf[seg_off[rank] : +seg_sizes[rank]]; one send and one receive per peer are posted as ONE group
    (ncclGroupStart/End under torch.distributed.batch_isend_irecv), so every pair talks over its own xGMI link at the same
    time instead of a ring moving padded segments.  Returns the list of outstanding works (empty when `host_staged`:
    gloo, the 1-GPU / CPU validation backend, moves host tensors and is waited for here)."""
    mine = gather_buf[seg_off[rank]:seg_off[rank] + seg_sizes[rank]]
    if world == 1:
        return []
    if host_staged:
        import torch
        src = mine.cpu()
        recv, ops = {}, []
        for d in range(1, world):
            to, frm = (rank + d) % world, (rank - d) % world
            recv[frm] = torch.zeros(seg_sizes[frm], dtype=torch.uint8)
            ops.append(dist.P2POp(dist.isend, src, to))
            ops.append(dist.P2POp(dist.irecv, recv[frm], frm))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        for frm, t in recv.items():
            gather_buf[seg_off[frm]:seg_off[frm] + seg_sizes[frm]].copy_(t)
        return []
    ops = []
    for d in range(1, world):
        to, frm = (rank + d) % world, (rank - d) % world
        ops.append(dist.P2POp(dist.isend, mine, to))
        ops.append(dist.P2POp(dist.irecv, gather_buf[seg_off[frm]:seg_off[frm] + seg_sizes[frm]], frm))
    return dist.batch_isend_irecv(ops)
