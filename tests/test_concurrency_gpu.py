"""The one-shot entry points are re-entrant per call (reference README: only optOff / optLevel are process-wide): eight host
threads call LZ4_compress_default / snappy_compress / ZSTD_compress / aocl_llc_compress (one GPU chunk and the threaded RAP
form) and the matching decoders in a loop on their own inputs; every output must equal the real reference's
(csrc/capi.hip context pool and scratch pool, csrc/rap.hip buffer cache)."""
import ctypes as C
import os
import sys
import threading

import pytest

import aocl_compression_amd as a
import oracle_lib as ol
from test_native_api import lib  # noqa: F401  (the bound native entry points)

pytestmark = pytest.mark.gpu

THREADS, ROUNDS = 8, 6
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_eight_threads_through_the_one_shot_apis(lib, eng, ref):  # noqa: F811
    kinds = list(a.corpus.KINDS)
    inputs = [a.corpus.make(kinds[t % len(kinds)], 60000 + 9000 * t, seed=100 + t).tobytes() for t in range(THREADS)]
    big = [a.corpus.make(kinds[(t + 2) % len(kinds)], (1 << 20) + 4096 * t + 17, seed=200 + t).tobytes() for t in range(THREADS)]
    want = [{"lz4": ref.compress(ol.LZ4, d), "snappy": ref.compress(ol.SNAPPY, d), "zstd": ref.compress(ol.ZSTD, d)} for d in inputs]
    errors = []
    lib.aocl_get_rap_frame_bound_mt.restype = C.c_int

    def worker(t):
        try:
            d = inputs[t]; n = len(d); w = want[t]
            for r in range(ROUNDS):
                out = C.create_string_buffer(n + n // 4 + 1024)
                back = C.create_string_buffer(n)
                k = lib.LZ4_compress_default(d, out, n, len(out))
                assert out.raw[:k] == w["lz4"], ("lz4", t, r)
                assert lib.LZ4_decompress_safe(out, back, k, n) == n and back.raw == d
                m = C.c_size_t(len(out))
                assert lib.snappy_compress(d, n, out, C.byref(m)) == 0 and out.raw[:m.value] == w["snappy"], ("snappy", t, r)
                m2 = C.c_size_t(n)
                assert lib.snappy_uncompress(out, m.value, back, C.byref(m2)) == 0 and back.raw[:m2.value] == d
                z = lib.ZSTD_compress(out, len(out), d, n, 1)
                assert not lib.ZSTD_isError(z) and out.raw[:z] == w["zstd"], ("zstd", t, r)
                assert lib.ZSTD_decompress(back, n, out, z) == n and back.raw == d
                # the unified API: one chunk, and the threaded form (a RAP stream of GPU partitions) on a larger input
                rr, blob, _ = eng.llc_compress(a.LZ4, d, n + n // 4 + 1024)
                assert rr > 0 and blob == w["lz4"], ("llc", t, r)
                b = big[t]
                cap = len(b) + len(b) // 4 + int(lib.aocl_get_rap_frame_bound_mt()) + 4096
                rr, blob, _ = eng.llc_compress(a.LZ4 if r % 2 else a.SNAPPY, b, cap, num_threads=4)
                assert rr > 0, ("llc threads", t, r)
                r2, plain, _ = eng.llc_decompress(a.LZ4 if r % 2 else a.SNAPPY, blob, len(b))
                assert r2 == len(b) and plain == b, ("llc threads round trip", t, r)
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(THREADS)]
    for th in ths: th.start()
    for th in ths: th.join()
    assert not errors, errors


def test_two_batches_in_flight_on_two_streams(eng):  # noqa: F811
    """bench.py's `two_batches_in_flight` leg at a small size: LZ4 batches on two HIP streams with their own output buffers overlap on
    the chip and both decode back to the source (LZ4 keeps all its state in LDS; Snappy's and Zstd's workspaces are per stream)."""
    import types
    sys.path.insert(0, ROOT)
    import bench
    args = types.SimpleNamespace(gib=0.25, chunk=0)
    r = bench.two_batches_in_flight(eng, args, a.corpus.silesia_mix(8 << 20, seed=5), steps=4)
    assert r["round_trip"] is True and r["two_streams_ms_per_step"] > 0 and r["one_stream_ms_per_step"] > 0
