"""ctypes binding of libaocl_compression_gpu.so (include/aocl_gpu_codec.h, include/aocl_compression.h).

Host-side mirror of the reference's interfaces for the hot path:
  * Engine.compress_chunks / decompress_chunks  <->  the per-codec native one-shot calls applied to every
    chunk of a batch (LZ4_compress_default / snappy::RawCompress / ZSTD_compress and their inverses);
  * Engine.llc_compress / llc_decompress        <->  aocl_llc_compress / aocl_llc_decompress
    (api/aocl_compression.h:170-207) on host buffers.
PyTorch is used only as the owner of device memory and streams; every compute call goes through the
C-ABI.  There is no CPU fallback: if the shared library is missing, or no gfx950 device is visible,
construction raises EngineError.
"""
from __future__ import annotations

import ctypes as C
import os

LZ4, SNAPPY, ZSTD = 0, 1, 2
_NAMES = {LZ4: "lz4", SNAPPY: "snappy", ZSTD: "zstd"}
# aocl_compression_type values (api/aocl_compression.h:109-119)
_LLC_ENUM = {LZ4: 0, SNAPPY: 4, ZSTD: 6}

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path() -> str:
    # AOCLGPU_LIB: development hook of tools/ (experiment / profiling builds made by build.py --variant); tests and bench.py
    # run without it and load the product library
    return os.environ.get("AOCLGPU_LIB") or os.path.join(_HERE, "libaocl_compression_gpu.so")


class EngineError(RuntimeError):
    pass


class Batch(C.Structure):
    """aoclgpu_batch (include/aocl_gpu_codec.h)."""
    _fields_ = [
        ("src", C.c_void_p), ("dst", C.c_void_p),
        ("src_off", C.c_void_p), ("src_len", C.c_void_p), ("dst_off", C.c_void_p), ("dst_cap", C.c_void_p),
        ("out_len", C.c_void_p),
        ("nchunks", C.c_uint32), ("reserved", C.c_uint32),
        ("total_bytes", C.c_uint64), ("chunk_bytes", C.c_uint64), ("slot_bytes", C.c_uint64),
        ("aux", C.c_void_p),
    ]


FLAG_RAP_PARTITIONS = 0x100
FLAG_ZSTD_LEVEL2 = 0x400      # Zstd level 2 instead of 1 (include/aocl_gpu_codec.h)
# kernel selection (never changes an output byte; lets tests drive every kernel on small batches)
FLAG_SNAPPY_TABLES_LDS, FLAG_SNAPPY_TABLES_HBM, FLAG_SNAPPY_TABLES_MIXED, FLAG_LZ4_TABLE_U32 = 0x1000, 0x2000, 0x4000, 0x0800


FLAG_ZSTD_UPSTREAM_TABLE = 0x80000


def zstd_level(level: int, upstream_table: bool = False) -> int:
    """AOCLGPU_ZSTD_LEVEL(level) [| AOCLGPU_FLAG_ZSTD_UPSTREAM_TABLE]: Zstd level 1..4 as batch flags"""
    return ((int(level) & 7) << 16) | (FLAG_ZSTD_UPSTREAM_TABLE if upstream_table else 0)


def lz4_acceleration(a: int) -> int:
    """AOCLGPU_LZ4_ACCELERATION(a): LZ4_compress_fast's acceleration (1..65537) as batch flags"""
    return (min(max(int(a), 1), 65537) - 1) << 15



class ZstdFrames(C.Structure):
    """aoclgpu_zstd_frames (include/aocl_gpu_codec.h)"""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("src_off", C.c_void_p), ("dst_off", C.c_void_p), ("out_len", C.c_void_p),
                ("nframes", C.c_uint32), ("windowLog", C.c_uint32), ("hashLog", C.c_uint32), ("minMatch", C.c_uint32), ("targetLength", C.c_uint32),
                ("strategy", C.c_uint32), ("chainLog", C.c_uint32), ("flags", C.c_uint32)]


class LlcDesc(C.Structure):
    """aocl_compression_desc (api/aocl_compression.h:125-152)."""
    _fields_ = [
        ("inBuf", C.c_void_p), ("outBuf", C.c_void_p), ("workBuf", C.c_void_p),
        ("inSize", C.c_size_t), ("outSize", C.c_size_t), ("level", C.c_size_t), ("optVar", C.c_size_t),
        ("numThreads", C.c_int), ("numMPIranks", C.c_int), ("memLimit", C.c_size_t), ("measureStats", C.c_int),
        ("cSize", C.c_uint64), ("dSize", C.c_uint64), ("cTime", C.c_uint64), ("dTime", C.c_uint64),
        ("cSpeed", C.c_float), ("dSpeed", C.c_float), ("optOff", C.c_int), ("optLevel", C.c_int),
    ]


EXPORTS = [
    "aoclgpu_device_count", "aoclgpu_init", "aoclgpu_last_error", "aoclgpu_version", "aoclgpu_compress_bound",
    "aoclgpu_lz4_compress_chunks", "aoclgpu_lz4_decompress_chunks", "aoclgpu_lz4_error_value",
    "aoclgpu_snappy_compress_chunks", "aoclgpu_snappy_decompress_chunks",
    "aoclgpu_zstd_compress_chunks", "aoclgpu_zstd_decompress_chunks", "aoclgpu_zstd_last_split_slabs",
    "aoclgpu_compact_slots", "aoclgpu_time_batch", "aoclgpu_compress_host", "aoclgpu_compress_host_ex", "aoclgpu_decompress_host",
    "aoclgpu_rap_pack", "aoclgpu_rap_unpack", "aoclgpu_rap_frame_bytes", "aoclgpu_rap_default_chunk",
    "aoclgpu_rap_compress_host", "aoclgpu_rap_decompress_host",
    "aoclgpu_zstd_compress_host_frames", "aoclgpu_zstd_decompress_host_frames", "aoclgpu_set_host_device", "aoclgpu_host_device",
    "aoclgpu_rap_assemble", "aoclgpu_zstd_compress_frames", "aoclgpu_zstd_compress_host_frame",
    "aoclgpu_zstd_set_large_input_mode", "aoclgpu_zstd_large_input_mode", "aoclgpu_lz4f_set_block_linking", "aoclgpu_lz4f_block_linking",
    "aocl_llc_compress", "aocl_llc_decompress", "aocl_llc_setup", "aocl_llc_destroy", "aocl_llc_version",
    "aoclgpu_probe_copy", "aoclgpu_probe_copy_variant", "aoclgpu_probe_scatter8", "aoclgpu_host_trim", "aoclgpu_host_cache_bytes",
    "aoclgpu_rap_compress_host_ex", "aoclgpu_zstd_rap_compress_host_parts", "aoclgpu_rap_assemble_slabs", "aoclgpu_set_rap_geometry", "aoclgpu_rap_geometry",
    "aoclgpu_comm_unique_id", "aoclgpu_comm_create", "aoclgpu_comm_destroy", "aoclgpu_comm_count", "aoclgpu_exchange_segments", "aoclgpu_allgather_words",
]


def load_library() -> C.CDLL:
    p = lib_path()
    if not os.path.exists(p):
        raise EngineError(f"{p} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (no CPU fallback)")
    lib = C.CDLL(p)
    lib.aoclgpu_last_error.restype = C.c_char_p
    lib.aoclgpu_version.restype = C.c_char_p
    lib.aocl_llc_version.restype = C.c_char_p
    lib.aoclgpu_compress_bound.restype = C.c_size_t
    lib.aoclgpu_compress_bound.argtypes = [C.c_int, C.c_size_t]
    for n in EXPORTS:
        if n.endswith("_chunks"):
            f = getattr(lib, n)
            f.restype = C.c_int
            f.argtypes = [C.POINTER(Batch), C.c_void_p]
    lib.aoclgpu_compact_slots.restype = C.c_int
    lib.aoclgpu_compact_slots.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.aoclgpu_time_batch.restype = C.c_double
    lib.aoclgpu_host_cache_bytes.restype = C.c_uint64
    lib.aoclgpu_comm_unique_id.argtypes = [C.c_void_p]
    lib.aoclgpu_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    lib.aoclgpu_comm_destroy.argtypes = [C.c_void_p]
    lib.aoclgpu_comm_count.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.aoclgpu_exchange_segments.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.aoclgpu_allgather_words.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
    lib.aoclgpu_rap_assemble_slabs.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                               C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p]
    lib.aoclgpu_rap_compress_host_ex.restype = C.c_int64
    lib.aoclgpu_rap_compress_host_ex.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32]
    lib.aoclgpu_zstd_rap_compress_host_parts.restype = C.c_int64
    lib.aoclgpu_zstd_rap_compress_host_parts.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t] + [C.c_uint32] * 7
    lib.aoclgpu_probe_copy.restype = C.c_double
    lib.aoclgpu_probe_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    lib.aoclgpu_probe_scatter8.restype = C.c_double
    lib.aoclgpu_probe_scatter8.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_uint64), C.c_void_p]
    lib.aoclgpu_time_batch.argtypes = [C.c_int, C.c_int, C.POINTER(Batch), C.c_int, C.c_void_p]
    for n in ("aoclgpu_compress_host", "aoclgpu_decompress_host"):
        f = getattr(lib, n)
        f.restype = C.c_int64
        f.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    lib.aoclgpu_compress_host_ex.restype = C.c_int64
    lib.aoclgpu_compress_host_ex.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint32]
    lib.aoclgpu_rap_pack.restype = C.c_int
    lib.aoclgpu_rap_pack.argtypes = [C.c_int, C.POINTER(Batch), C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p]
    lib.aoclgpu_rap_unpack.restype = C.c_int
    lib.aoclgpu_rap_unpack.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_int64), C.c_void_p]
    lib.aoclgpu_rap_assemble.restype = C.c_int
    lib.aoclgpu_rap_assemble.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                         C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p]
    for n in ("aoclgpu_zstd_compress_host_frames",):
        f = getattr(lib, n)
        f.restype = C.c_int64
        f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint32]
    lib.aoclgpu_zstd_decompress_host_frames.restype = C.c_int64
    lib.aoclgpu_zstd_decompress_host_frames.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    lib.aoclgpu_zstd_compress_frames.restype = C.c_int
    lib.aoclgpu_zstd_compress_frames.argtypes = [C.POINTER(ZstdFrames), C.c_void_p]
    lib.aoclgpu_zstd_compress_host_frame.restype = C.c_int64
    lib.aoclgpu_zstd_compress_host_frame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t] + [C.c_uint32] * 7
    lib.aoclgpu_zstd_set_large_input_mode.restype = None
    lib.aoclgpu_zstd_set_large_input_mode.argtypes = [C.c_int]
    lib.aoclgpu_rap_frame_bytes.restype = C.c_uint64
    lib.aoclgpu_rap_frame_bytes.argtypes = [C.c_int, C.c_uint32]
    for n in ("aocl_llc_compress", "aocl_llc_decompress"):
        f = getattr(lib, n)
        f.restype = C.c_int64
        f.argtypes = [C.POINTER(LlcDesc), C.c_int]
    lib.aocl_llc_setup.restype = C.c_int32
    lib.aocl_llc_setup.argtypes = [C.POINTER(LlcDesc), C.c_int]
    lib.aocl_llc_destroy.restype = None
    lib.aocl_llc_destroy.argtypes = [C.POINTER(LlcDesc), C.c_int]
    return lib


def compress_bound(codec: int, n: int) -> int:
    """LZ4_COMPRESSBOUND / snappy MaxCompressedLength / ZSTD_COMPRESSBOUND (pure arithmetic, no GPU)."""
    if codec == LZ4:
        return 0 if n > 0x7E000000 else n + n // 255 + 16
    if codec == SNAPPY:
        return 32 + n + n // 6
    return n + (n >> 8) + (((128 << 10) - n) >> 11 if n < (128 << 10) else 0)


class Engine:
    """One process / one GPU.  All tensors are torch uint8 / int32 tensors on that GPU."""

    def __init__(self, device: int = 0):
        import torch  # device memory + streams only
        self.torch = torch
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise EngineError("no GPU visible: the engine has no CPU fallback")
        rc = self.lib.aoclgpu_init(device)
        if rc != 0:
            raise EngineError(f"aoclgpu_init({device}) failed: {self.lib.aoclgpu_last_error().decode()}")
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.force_flags = 0      # AOCLGPU_FLAG_* kernel-selection bits OR-ed into every compress call (test hook)

    # ---- helpers ----
    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _fn(self, codec: int, op: str):
        return getattr(self.lib, f"aoclgpu_{_NAMES[codec]}_{op}_chunks")

    def _check(self, rc, what):
        if rc != 0:
            raise EngineError(f"{what} failed ({rc}): {self.lib.aoclgpu_last_error().decode()}")

    def uniform_compress_batch(self, src, chunk_bytes: int, slots, out_len, slot_bytes: int) -> Batch:
        n = int(src.numel())
        nchunks = max(1, -(-n // chunk_bytes))
        assert slots.numel() >= nchunks * slot_bytes and out_len.numel() >= nchunks
        return Batch(src=src.data_ptr(), dst=slots.data_ptr(), out_len=out_len.data_ptr(), nchunks=nchunks,
                     total_bytes=n, chunk_bytes=chunk_bytes, slot_bytes=slot_bytes, reserved=self.force_flags)

    def uniform_decompress_batch(self, slots, comp_len, slot_bytes: int, dst, chunk_bytes: int, out_len) -> Batch:
        n = int(dst.numel())
        nchunks = max(1, -(-n // chunk_bytes))
        assert comp_len.numel() >= nchunks and out_len.numel() >= nchunks
        return Batch(src=slots.data_ptr(), dst=dst.data_ptr(), src_len=comp_len.data_ptr(), out_len=out_len.data_ptr(),
                     nchunks=nchunks, total_bytes=n, chunk_bytes=chunk_bytes, slot_bytes=slot_bytes)

    # ---- batch API on device tensors ----
    def compress_chunks(self, codec: int, src, chunk_bytes: int, slot_bytes: int | None = None, flags: int = 0):
        """Compress every chunk_bytes-sized chunk of `src` independently (flags: AOCLGPU_FLAG_*, e.g. FLAG_ZSTD_LEVEL2).
        Returns (slots uint8[nchunks*slot_bytes], out_len int32[nchunks], slot_bytes)."""
        t = self.torch
        if slot_bytes is None:
            slot_bytes = (compress_bound(codec, chunk_bytes) + 15) & ~15
        nchunks = max(1, -(-int(src.numel()) // chunk_bytes))
        slots = t.empty(nchunks * slot_bytes, dtype=t.uint8, device=self.device)
        out_len = t.zeros(nchunks, dtype=t.int32, device=self.device)
        b = self.uniform_compress_batch(src, chunk_bytes, slots, out_len, slot_bytes)
        b.reserved |= flags
        self._check(self._fn(codec, "compress")(C.byref(b), self._stream()), f"{_NAMES[codec]} compress")
        return slots, out_len, slot_bytes

    def decompress_chunks(self, codec: int, slots, comp_len, slot_bytes: int, total_bytes: int, chunk_bytes: int):
        """Inverse of compress_chunks.  Returns (dst uint8[total_bytes], out_len int32[nchunks])."""
        t = self.torch
        nchunks = max(1, -(-total_bytes // chunk_bytes))
        dst = t.empty(total_bytes, dtype=t.uint8, device=self.device)
        out_len = t.zeros(nchunks, dtype=t.int32, device=self.device)
        comp_u32 = comp_len.to(t.int32).contiguous()
        b = self.uniform_decompress_batch(slots, comp_u32, slot_bytes, dst, chunk_bytes, out_len)
        self._check(self._fn(codec, "decompress")(C.byref(b), self._stream()), f"{_NAMES[codec]} decompress")
        return dst, out_len

    # ---- RAP streams (the reference's multi-threaded container) ----
    def rap_compress(self, codec: int, src, chunk_bytes: int):
        """Compress `src` in chunk_bytes partitions and join them into one RAP stream (aoclgpu_rap_pack).
        Returns the stream as a uint8 tensor."""
        t = self.torch
        slot_bytes = (compress_bound(codec, chunk_bytes) + 15) & ~15
        nchunks = max(1, -(-int(src.numel()) // chunk_bytes))
        slots = t.empty(nchunks * slot_bytes, dtype=t.uint8, device=self.device)
        out_len = t.zeros(nchunks, dtype=t.int32, device=self.device)
        aux = t.zeros(nchunks, dtype=t.int32, device=self.device)
        b = self.uniform_compress_batch(src, chunk_bytes, slots, out_len, slot_bytes)
        b.aux = aux.data_ptr()
        self._check(self._fn(codec, "compress")(C.byref(b), self._stream()), f"{_NAMES[codec]} compress")
        cap = int(self.lib.aoclgpu_rap_frame_bytes(codec, nchunks)) + nchunks * slot_bytes + 16
        out = t.empty(cap, dtype=t.uint8, device=self.device)
        n = C.c_uint64(0)
        self._check(self.lib.aoclgpu_rap_pack(codec, C.byref(b), out.data_ptr(), cap, C.byref(n), self._stream()), "rap_pack")
        return out[:n.value]

    def zstd_rap_compress(self, src, part_off, window_log: int, hash_log: int, min_match: int, strategy: int = 1, chain_log: int = 0,
                          target_length: int = 0):
        """Zstd RAP stream with partitions of any size: partition i covers src[part_off[i], part_off[i+1]) and becomes one frame
        with the given compression parameters (what every thread of the reference's AOCL_ENABLE_THREADS build does with the
        caller's parameters, zstd_compress.c:5447-5468), joined behind the RAP frame (aoclgpu_rap_assemble)."""
        dst, doff, lens = self.zstd_compress_frames(src, part_off, window_log, hash_log, min_match, target_length=target_length,
                                                    strategy=strategy, chain_log=chain_log)
        if any(x <= 0 for x in lens):
            raise EngineError("zstd_rap_compress: a partition failed to compress")
        sizes = [int(part_off[i + 1]) - int(part_off[i]) for i in range(len(lens))]
        if len(lens) == 1:
            return dst[:lens[0]]
        return self.rap_assemble(ZSTD, dst, doff[:-1], lens, sizes)

    def rap_decompress(self, codec: int, stream, max_bytes: int):
        """Decode a RAP stream (or a plain one-shot stream).  Returns (dst tensor, decoded size or negative)."""
        t = self.torch
        dst = t.empty(max(max_bytes, 1), dtype=t.uint8, device=self.device)
        r = C.c_int64(0)
        self._check(self.lib.aoclgpu_rap_unpack(codec, stream.data_ptr(), int(stream.numel()), dst.data_ptr(), max_bytes,
                                                C.byref(r), self._stream()), "rap_unpack")
        return dst, r.value

    def rap_assemble(self, codec: int, gathered, chunk_off, chunk_len, chunk_src_len, chunk_aux=None):
        """ONE RAP stream from gathered one-shot chunk streams (aoclgpu_rap_assemble): `gathered` is a device tensor, the
        per-chunk arrays are host sequences in stream order.  Returns the stream as a uint8 tensor."""
        import numpy as np
        t = self.torch
        k = len(chunk_len)
        off = np.ascontiguousarray(chunk_off, dtype=np.uint64)
        ln = np.ascontiguousarray(chunk_len, dtype=np.int32)
        sl = np.ascontiguousarray(chunk_src_len, dtype=np.uint32)
        ax = np.ascontiguousarray(chunk_aux, dtype=np.uint32) if chunk_aux is not None else None
        cap = int(self.lib.aoclgpu_rap_frame_bytes(codec, k)) + int(ln.clip(min=0).sum()) + 16 * k + 64
        out = t.empty(cap, dtype=t.uint8, device=self.device)
        n = C.c_uint64(0)
        self._check(self.lib.aoclgpu_rap_assemble(codec, gathered.data_ptr(), off.ctypes.data, ln.ctypes.data, sl.ctypes.data,
                                                  ax.ctypes.data if ax is not None else None, k, out.data_ptr(), cap,
                                                  C.byref(n), self._stream()), "rap_assemble")
        return out[:n.value]

    def rap_assemble_slabs(self, codec: int, gathered, chunk_off, chunk_len, chunk_src_len, chunk_aux=None, slab_src_bytes: int = 0):
        """A stream of any length as the sequence of RAP streams the container allows (aoclgpu_rap_assemble_slabs).
        Returns (device tensor holding the slab streams back to back, slab offsets (nslabs + 1), first chunk of each slab)."""
        import numpy as np
        t = self.torch
        k = len(chunk_len)
        off = np.ascontiguousarray(chunk_off, dtype=np.uint64)
        ln = np.ascontiguousarray(chunk_len, dtype=np.int32)
        sl = np.ascontiguousarray(chunk_src_len, dtype=np.uint32)
        ax = np.ascontiguousarray(chunk_aux, dtype=np.uint32) if chunk_aux is not None else None
        slab = slab_src_bytes or (1 << 30)
        max_slabs = int(sl.astype(np.uint64).sum() // max(1, slab - int(sl.max()))) + k // 65535 + 4
        cap = int(ln.clip(min=0).astype(np.int64).sum()) + 16 * k + (64 + int(self.lib.aoclgpu_rap_frame_bytes(codec, 65535))) * max_slabs
        out = t.empty(cap, dtype=t.uint8, device=self.device)
        so = np.zeros(max_slabs + 1, np.uint64)
        sf = np.zeros(max_slabs + 1, np.uint64)
        ns = C.c_uint32(0)
        self._check(self.lib.aoclgpu_rap_assemble_slabs(codec, gathered.data_ptr(), off.ctypes.data, ln.ctypes.data, sl.ctypes.data,
                                                        ax.ctypes.data if ax is not None else None, k, slab, out.data_ptr(), cap,
                                                        so.ctypes.data, sf.ctypes.data, max_slabs, C.byref(ns), self._stream()), "rap_assemble_slabs")
        n = ns.value
        return out[:int(so[n])], [int(x) for x in so[:n + 1]], [int(x) for x in sf[:n + 1]]

    def zstd_compress_frames(self, src, src_off, window_log: int, hash_log: int, min_match: int, caps=None, target_length: int = 0,
                             strategy: int = 1, chain_log: int = 0, flags: int = 0):
        """One Zstd frame per input of any size (aoclgpu_zstd_compress_frames): `src` is a device uint8 tensor, `src_off` a
        host sequence of nframes + 1 offsets into it.  Returns (dst tensor, dst offsets as a list, frame sizes as a list)."""
        import numpy as np
        t = self.torch
        so = np.ascontiguousarray(src_off, dtype=np.uint64)
        k = len(so) - 1
        sizes = (so[1:] - so[:-1]).astype(np.int64)
        if caps is None:
            caps = [compress_bound(ZSTD, int(n)) + 32 for n in sizes]
        do = np.zeros(k + 1, dtype=np.uint64)
        do[1:] = np.cumsum(np.asarray(caps, dtype=np.uint64))
        dst = t.empty(max(int(do[-1]), 1), dtype=t.uint8, device=self.device)
        d_so = t.from_numpy(so.view(np.int64)).to(self.device)
        d_do = t.from_numpy(do.view(np.int64)).to(self.device)
        out_len = t.zeros(max(k, 1), dtype=t.int64, device=self.device)
        f = ZstdFrames(src.data_ptr(), dst.data_ptr(), d_so.data_ptr(), d_do.data_ptr(), out_len.data_ptr(), k,
                       window_log, hash_log, min_match, target_length, strategy, chain_log, flags)
        self._check(self.lib.aoclgpu_zstd_compress_frames(C.byref(f), self._stream()), "zstd_compress_frames")
        return dst, [int(x) for x in do], [int(x) for x in out_len[:k].cpu().tolist()]

    def compact(self, slots, slot_bytes: int, out_len):
        """Gather the used prefix of each slot into one stream; returns (packed, offsets uint64-as-int64[n+1])."""
        t = self.torch
        n = int(out_len.numel())
        total = int(out_len.clamp(min=0).sum().item())
        packed = t.empty(max(total, 1), dtype=t.uint8, device=self.device)
        off = t.zeros(n + 1, dtype=t.int64, device=self.device)
        self._check(self.lib.aoclgpu_compact_slots(slots.data_ptr(), slot_bytes, out_len.data_ptr(), n, packed.data_ptr(),
                                                   off.data_ptr(), self._stream()), "compact")
        return packed[:total], off

    def time_batch(self, codec: int, op: int, batch: Batch, iters: int) -> float:
        ms = self.lib.aoclgpu_time_batch(codec, op, C.byref(batch), iters, self._stream())
        if ms < 0:
            raise EngineError(f"time_batch failed: {self.lib.aoclgpu_last_error().decode()}")
        return ms

    # ---- host one-shot (native-API contract) ----
    def compress_host(self, codec: int, data: bytes, cap: int | None = None, flags: int = 0) -> bytes | None:
        cap = compress_bound(codec, len(data)) if cap is None else cap
        out = C.create_string_buffer(max(cap, 1))
        r = self.lib.aoclgpu_compress_host_ex(codec, data, len(data), out, cap, flags | self.force_flags)
        if r < 0:
            raise EngineError(f"compress_host failed ({r}): {self.lib.aoclgpu_last_error().decode()}")
        return out.raw[:r] if r > 0 else None

    def decompress_host(self, codec: int, data: bytes, cap: int) -> bytes | None:
        out = C.create_string_buffer(max(cap, 1))
        r = self.lib.aoclgpu_decompress_host(codec, data, len(data), out, cap)
        if r <= -(1 << 39):
            raise EngineError(f"decompress_host failed ({r}): {self.lib.aoclgpu_last_error().decode()}")
        return out.raw[:r] if r >= 0 else None

    # ---- unified API (aocl_llc_*) on host buffers ----
    def llc_compress(self, codec: int, data: bytes, cap: int, level: int = 1, stats: bool = False, num_threads: int = 0, opt_var: int = 0):
        out = C.create_string_buffer(max(cap, 1))
        inb = C.create_string_buffer(data, len(data)) if data else None
        d = LlcDesc(inBuf=C.cast(inb, C.c_void_p) if inb else None, outBuf=C.cast(out, C.c_void_p), inSize=len(data),
                    outSize=cap, level=level, measureStats=1 if stats else 0, numThreads=num_threads, optVar=opt_var)
        r = self.lib.aocl_llc_compress(C.byref(d), _LLC_ENUM[codec])
        return r, (out.raw[:r] if r > 0 else b""), d

    def llc_decompress(self, codec: int, data: bytes, cap: int, stats: bool = False):
        out = C.create_string_buffer(max(cap, 1))
        inb = C.create_string_buffer(data, len(data)) if data else None
        d = LlcDesc(inBuf=C.cast(inb, C.c_void_p) if inb else None, outBuf=C.cast(out, C.c_void_p), inSize=len(data),
                    outSize=cap, measureStats=1 if stats else 0)
        r = self.lib.aocl_llc_decompress(C.byref(d), _LLC_ENUM[codec])
        return r, (out.raw[:r] if r > 0 else b""), d


_ENGINE = None


def engine(device: int = 0) -> Engine:
    global _ENGINE
    if _ENGINE is None:
        _ENGINE = Engine(device)
    return _ENGINE
