"""Chunk sharding across ranks (the GPU counterpart of the reference's RAP partition split,
threads/threads.c:57-80,121-153): contiguous chunk ranges, ordered gather of compressed segments."""
from __future__ import annotations


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
