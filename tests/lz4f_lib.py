"""ctypes view of the LZ4 frame API (lz4frame.h), shared by the GPU library and the reference build.  TEST INFRASTRUCTURE."""
import ctypes as C


class FrameInfo(C.Structure):
    _fields_ = [("blockSizeID", C.c_int), ("blockMode", C.c_int), ("contentChecksumFlag", C.c_int), ("frameType", C.c_int),
                ("contentSize", C.c_ulonglong), ("dictID", C.c_uint), ("blockChecksumFlag", C.c_int)]


class Prefs(C.Structure):
    _fields_ = [("frameInfo", FrameInfo), ("compressionLevel", C.c_int), ("autoFlush", C.c_uint), ("favorDecSpeed", C.c_uint),
                ("reserved", C.c_uint * 3)]


def prefs(bsid=0, independent=False, content_checksum=False, block_checksum=False, content_size=False, level=0, auto_flush=False):
    p = Prefs()
    p.frameInfo.blockSizeID = bsid
    p.frameInfo.blockMode = 1 if independent else 0
    p.frameInfo.contentChecksumFlag = int(content_checksum)
    p.frameInfo.blockChecksumFlag = int(block_checksum)
    p.frameInfo.contentSize = 1 if content_size else 0
    p.compressionLevel = level
    p.autoFlush = int(auto_flush)
    return p


def bind(L):
    L.LZ4F_isError.restype = C.c_uint
    L.LZ4F_isError.argtypes = [C.c_size_t]
    L.LZ4F_getErrorName.restype = C.c_char_p
    L.LZ4F_getErrorName.argtypes = [C.c_size_t]
    L.LZ4F_compressFrameBound.restype = C.c_size_t
    L.LZ4F_compressFrameBound.argtypes = [C.c_size_t, C.POINTER(Prefs)]
    L.LZ4F_compressFrame.restype = C.c_size_t
    L.LZ4F_compressFrame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(Prefs)]
    L.LZ4F_createDecompressionContext.restype = C.c_size_t
    L.LZ4F_createDecompressionContext.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    L.LZ4F_freeDecompressionContext.restype = C.c_size_t
    L.LZ4F_freeDecompressionContext.argtypes = [C.c_void_p]
    L.LZ4F_decompress.restype = C.c_size_t
    L.LZ4F_decompress.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p]
    L.LZ4F_getFrameInfo.restype = C.c_size_t
    L.LZ4F_getFrameInfo.argtypes = [C.c_void_p, C.POINTER(FrameInfo), C.c_void_p, C.POINTER(C.c_size_t)]
    L.LZ4F_headerSize.restype = C.c_size_t
    L.LZ4F_headerSize.argtypes = [C.c_void_p, C.c_size_t]
    L.LZ4F_createCompressionContext.restype = C.c_size_t
    L.LZ4F_createCompressionContext.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    L.LZ4F_freeCompressionContext.restype = C.c_size_t
    L.LZ4F_freeCompressionContext.argtypes = [C.c_void_p]
    L.LZ4F_compressBound.restype = C.c_size_t
    L.LZ4F_compressBound.argtypes = [C.c_size_t, C.POINTER(Prefs)]
    L.LZ4F_compressBegin.restype = C.c_size_t
    L.LZ4F_compressBegin.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(Prefs)]
    L.LZ4F_compressUpdate.restype = C.c_size_t
    L.LZ4F_compressUpdate.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    L.LZ4F_flush.restype = C.c_size_t
    L.LZ4F_flush.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.LZ4F_compressEnd.restype = C.c_size_t
    L.LZ4F_compressEnd.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    return L


def compress_stream(L, data: bytes, p, pieces, flush_at=()):
    """LZ4F_compressBegin / Update (one call per piece, LZ4F_flush after the pieces listed in flush_at) / End; frame bytes or error name"""
    ctx = C.c_void_p()
    assert not L.LZ4F_isError(L.LZ4F_createCompressionContext(C.byref(ctx), 100))
    try:
        out = bytearray()
        buf = C.create_string_buffer(64)
        r = L.LZ4F_compressBegin(ctx, buf, 64, C.byref(p))
        if L.LZ4F_isError(r):
            return L.LZ4F_getErrorName(r).decode()
        out += buf.raw[:r]
        pos = 0
        for k, n in enumerate(pieces):
            piece = data[pos:pos + n]; pos += n
            cap = L.LZ4F_compressBound(len(piece), C.byref(p))
            buf = C.create_string_buffer(cap + 1)
            r = L.LZ4F_compressUpdate(ctx, buf, cap, piece, len(piece), None)
            if L.LZ4F_isError(r):
                return L.LZ4F_getErrorName(r).decode()
            out += buf.raw[:r]
            if k in flush_at:
                r = L.LZ4F_flush(ctx, buf, cap, None)
                if L.LZ4F_isError(r):
                    return L.LZ4F_getErrorName(r).decode()
                out += buf.raw[:r]
        cap = L.LZ4F_compressBound(0, C.byref(p))
        buf = C.create_string_buffer(cap + 1)
        r = L.LZ4F_compressEnd(ctx, buf, cap, None)
        if L.LZ4F_isError(r):
            return L.LZ4F_getErrorName(r).decode()
        out += buf.raw[:r]
        return bytes(out)
    finally:
        L.LZ4F_freeCompressionContext(ctx)


def compress(L, data: bytes, p=None):
    """frame bytes, or the error name"""
    cap = L.LZ4F_compressFrameBound(len(data), C.byref(p) if p is not None else None)
    out = C.create_string_buffer(cap + 1)
    r = L.LZ4F_compressFrame(out, cap, data, len(data), C.byref(p) if p is not None else None)
    if L.LZ4F_isError(r):
        return L.LZ4F_getErrorName(r).decode()
    return out.raw[:r]


def decompress(L, frame: bytes, cap: int):
    """(decoded bytes, consumed) or the error name; the whole frame is handed over in one call"""
    ctx = C.c_void_p()
    assert not L.LZ4F_isError(L.LZ4F_createDecompressionContext(C.byref(ctx), 100))
    try:
        out = C.create_string_buffer(max(cap, 1))
        dn, sn = C.c_size_t(cap), C.c_size_t(len(frame))
        r = L.LZ4F_decompress(ctx, out, C.byref(dn), frame, C.byref(sn), None)
        if L.LZ4F_isError(r):
            return L.LZ4F_getErrorName(r).decode()
        if r != 0:
            return f"hint {r}"
        return out.raw[:dn.value], sn.value
    finally:
        L.LZ4F_freeDecompressionContext(ctx)
