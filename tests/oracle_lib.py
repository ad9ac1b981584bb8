"""ctypes access to the CPU checkers: oracle/liboracle.so (this repo's C restatement) and, when present,
oracle/_ref/libaocl_ref.so (the real reference compiled in place).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libaocl_ref.so")
REF_MT_SO = os.path.join(ORACLE_DIR, "_ref", "libaocl_ref_mt.so")   # the same sources with AOCL_ENABLE_THREADS

LZ4, SNAPPY, ZSTD = 0, 1, 2


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "oracle"], check=True)
    return ORACLE_SO


def bound(codec, n):
    if codec == LZ4:
        return n + n // 255 + 16
    if codec == SNAPPY:
        return 32 + n + n // 6
    return n + (n >> 8) + (((128 << 10) - n) >> 11 if n < (128 << 10) else 0)


class Oracle:
    def __init__(self):
        build_oracle()
        L = self.L = C.CDLL(ORACLE_SO)
        def sig(name, restype, argtypes=None):
            f = getattr(L, name, None)
            if f is None:       # codec restatement not built yet
                return
            f.restype = restype
            if argtypes is not None:
                f.argtypes = argtypes
        sig("orc_lz4_compress", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int])
        sig("orc_lz4_decompress", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int])
        sig("orc_snappy_compress", C.c_size_t, [C.c_void_p, C.c_size_t, C.c_void_p])
        sig("orc_snappy_max_compressed_length", C.c_size_t, [C.c_size_t])
        sig("orc_snappy_uncompress", C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t])
        sig("orc_snappy_uncompressed_length", C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)])
        sig("orc_zstd_compress_bound", C.c_size_t, [C.c_size_t])
        sig("orc_zstd_compress", C.c_size_t, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t])
        sig("orc_zstd_compress_level", C.c_size_t, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int])
        sig("orc_zstd_decompress", C.c_size_t, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t])
        sig("orc_compress_chunks", C.c_int,
            [C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int])
        sig("orc_decompress_chunks", C.c_int,
            [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int])

    def compress(self, codec, data: bytes, cap=None, level=1):
        """bytes or None (failure); level: Zstd only (1 or 2)"""
        n = len(data)
        cap = bound(codec, n) if cap is None else cap
        out = C.create_string_buffer(max(cap, 1))
        if codec == LZ4:
            r = self.L.orc_lz4_compress(data, n, out, cap)
        elif codec == SNAPPY:
            assert cap >= bound(SNAPPY, n)
            r = self.L.orc_snappy_compress(data, n, out)
        else:
            r = self.L.orc_zstd_compress_level(data, n, out, cap, level)
        return out.raw[:r] if r > 0 else None

    def decompress(self, codec, data: bytes, cap: int):
        out = C.create_string_buffer(max(cap, 1))
        if codec == LZ4:
            r = self.L.orc_lz4_decompress(data, len(data), out, cap)
            return out.raw[:r] if r >= 0 else None
        if codec == SNAPPY:
            ok = self.L.orc_snappy_uncompress(data, len(data), out, cap)
            if not ok:
                return None
            ln = C.c_size_t(0)
            self.L.orc_snappy_uncompressed_length(data, len(data), C.byref(ln))
            return out.raw[:ln.value]
        r = self.L.orc_zstd_decompress(data, len(data), out, cap)
        return None if r == C.c_size_t(-1).value else out.raw[:r]


class Ref:
    """The real reference (native one-shot APIs + aocl_llc_*)."""
    SO = REF_SO

    def __init__(self):
        if not os.path.exists(self.SO):
            if os.path.exists("/root/reference/api/api.cpp"):
                subprocess.run(["bash", os.path.join(ORACLE_DIR, "build_ref.sh")], check=True)
            else:
                raise FileNotFoundError(self.SO)
        L = self.L = C.CDLL(self.SO)
        L.ZSTD_compress.restype = C.c_size_t
        L.ZSTD_decompress.restype = C.c_size_t
        L.ZSTD_isError.restype = C.c_uint
        L.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
        L.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.ZSTD_isError.argtypes = [C.c_size_t]
        L.snappy_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t)]
        L.snappy_uncompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t)]

    def compress(self, codec, data: bytes, cap=None, level=1):
        n = len(data)
        cap = bound(codec, n) if cap is None else cap
        out = C.create_string_buffer(max(cap, 1))
        if codec == LZ4:
            r = self.L.LZ4_compress_default(data, out, n, cap)
            return out.raw[:r] if r > 0 else None
        if codec == SNAPPY:
            ln = C.c_size_t(cap)
            st = self.L.snappy_compress(data, n, out, C.byref(ln))
            return out.raw[:ln.value] if st == 0 else None
        r = self.L.ZSTD_compress(out, cap, data, n, level)
        return None if self.L.ZSTD_isError(r) else out.raw[:r]

    def decompress(self, codec, data: bytes, cap: int):
        out = C.create_string_buffer(max(cap, 1))
        if codec == LZ4:
            r = self.L.LZ4_decompress_safe(data, out, len(data), cap)
            return out.raw[:r] if r >= 0 else None
        if codec == SNAPPY:
            ln = C.c_size_t(cap)
            st = self.L.snappy_uncompress(data, len(data), out, C.byref(ln))
            return out.raw[:ln.value] if st == 0 else None
        r = self.L.ZSTD_decompress(out, cap, data, len(data))
        return None if self.L.ZSTD_isError(r) else out.raw[:r]


class RefMT(Ref):
    """The reference built with AOCL_ENABLE_THREADS: its one-shot functions write / read RAP streams
    (threads/threads.c, lz4.c:2655-2909, snappy.cc:2505-2654, zstd_compress.c:5343-5534)."""
    SO = REF_MT_SO

    def __init__(self):
        super().__init__()
        self.L.aocl_skip_rap_frame_mt.argtypes = [C.c_void_p, C.c_int32]
        self.gomp = C.CDLL("libgomp.so.1", mode=C.RTLD_GLOBAL)
        # the reference registers its threaded decoders in the codec's lazy native setup, which only the compress
        # entry points trigger (lz4.c: LZ4_decompress_wrapper_mt_fp starts as the single-threaded wrapper)
        for codec in (LZ4, SNAPPY, ZSTD):
            self.compress(codec, b"warm-up " * 8)

    def set_threads(self, n: int):
        self.gomp.omp_set_num_threads(int(n))

    def skip(self, stream: bytes) -> int:
        return self.L.aocl_skip_rap_frame_mt(stream, len(stream))

    def decompress(self, codec, data: bytes, cap: int):
        if codec != SNAPPY:
            return super().decompress(codec, data, cap)
        # the C binding sizes the output from the first bytes of the stream, which in a RAP stream is the frame;
        # the RAP-aware entry point is snappy::RawUncompress (snappy.cc:2271-2390)
        f = getattr(self.L, "_ZN6snappy13RawUncompressEPKcmPc")
        f.restype = C.c_bool
        f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        out = C.create_string_buffer(max(cap, 1))
        return out.raw[:cap] if f(data, len(data), out) else None


def have_ref_mt() -> bool:
    return os.path.exists(REF_MT_SO) or os.path.exists("/root/reference/api/api.cpp")


def have_ref() -> bool:
    return os.path.exists(REF_SO) or os.path.exists("/root/reference/api/api.cpp")
