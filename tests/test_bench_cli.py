"""The bench CLI (csrc/bench_cli.cpp, counterpart of the reference's aocl_compression_bench, test/codec_bench.c) and the
multi-partition mode of the unified API (numThreads > 1 -> RAP stream, as the reference's AOCL_ENABLE_THREADS build)."""
import ctypes as C
import os
import re
import subprocess

import pytest

import aocl_compression_amd as a
from aocl_compression_amd import api
import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "aocl-compression_amd", "aocl_compression_bench")


def _run(*args):
    r = subprocess.run([CLI, *args], capture_output=True, text=True, timeout=300)
    return r.returncode, r.stdout + r.stderr


def test_cli_help_and_method_list_run_without_a_gpu():
    assert os.path.exists(CLI), "run __graft_entry__.build() first"
    rc, out = _run("-l")
    assert rc == 0 and "LZ4\t\t NA\t\tNA" in out and "SNAPPY" in out and "ZSTD\t\t 1\t\t22" in out
    rc, out = _run("-h")
    assert rc == 0 and "Usage: aocl_compression_bench <options> input" in out and "-e<>:<>:<>" in out
    rc, out = _run("-exyz", "/dev/null")
    assert rc == 1 and "Unsupported option passed" in out
    rc, out = _run("-rdecompress", "/dev/null")
    assert rc == 1 and "Codec not specified" in out


@pytest.mark.gpu
@pytest.mark.parametrize("method,codec", [("lz4", ol.LZ4), ("snappy", ol.SNAPPY), ("zstd:1", ol.ZSTD)])
def test_cli_round_trip_dump_and_stats(eng, ref, ref_mt, tmp_path, method, codec):
    data = a.corpus.silesia_mix(3 * 1000 * 1000 + 17, seed=8, segment=200000).tobytes()
    f = tmp_path / "input.bin"
    f.write_bytes(data)
    rc, out = _run(f"-e{method}", "-t", "-p", "-i2", str(f))
    assert rc == 0, out
    assert "verification: passed" in out
    m = re.search(r"Compression:\s+speed\(avg\) ([\d.]+) MB/s, time\(avg\) ([\d.]+) ms, size (\d+),", out)
    assert m and "Decompression:" in out
    size = int(m.group(1) and m.group(3))
    assert f"Ratio:               {size * 100.0 / len(data):.2f}" in out
    # dump the compressed stream: a RAP stream any reference decoder reads
    dump = tmp_path / "out.cmp"
    rc, out = _run(f"-e{method}", "-rcompress", f"-d{dump}", "-i1", str(f))
    assert rc == 0, out
    stream = dump.read_bytes()
    assert len(stream) == size
    skip = ref_mt.skip(stream)
    assert ref.decompress(codec, stream[skip:], len(data)) == data          # legacy single-threaded decoder
    ref_mt.set_threads(4)
    assert ref_mt.decompress(codec, stream, len(data)) == data
    # -rdecompress of a stream written by the reference's threaded build, validated against the original
    ref_mt.set_threads(3)
    theirs = tmp_path / "ref.cmp"
    theirs.write_bytes(ref_mt.compress(codec, data, len(data) + len(data) // 6 + 4096))
    back = tmp_path / "back.bin"
    rc, out = _run(f"-e{method}", "-rdecompress", "-t", f"-f{f}", f"-d{back}", "-i1", str(theirs))
    assert rc == 0 and "verification: passed" in out, out
    assert back.read_bytes() == data


@pytest.mark.gpu
def test_cli_native_mode_is_the_one_shot_stream(eng, oracle, tmp_path):
    data = a.corpus.make("jsonlog", 100000, seed=4).tobytes()
    f = tmp_path / "small.bin"
    f.write_bytes(data)
    for method, codec in (("lz4", ol.LZ4), ("snappy", ol.SNAPPY), ("zstd:1", ol.ZSTD)):
        dump = tmp_path / f"{codec}.cmp"
        rc, out = _run(f"-e{method}", "-n", "-rcompress", f"-d{dump}", "-i1", str(f))
        assert rc == 0, out
        assert dump.read_bytes() == oracle.compress(codec, data)
    rc, out = _run("-ezstd:3", "-n", "-t", str(f))
    assert rc == 0 and "verification: passed" in out, out                    # double-fast
    rc, out = _run("-ezstd:7", "-n", "-t", str(f))
    assert rc == 1 and "Compression: failed" in out                          # levels 1-4 have a GPU backend
    rc, out = _run("-ezlib", "-t", str(f))
    assert rc == 1 and "no GPU backend" in out


@pytest.mark.gpu
def test_unified_api_with_threads_writes_and_reads_rap_streams(eng, ref, ref_mt):
    lib = api.load_library()
    data = a.corpus.silesia_mix(2 * 1000 * 1000, seed=9, segment=100000).tobytes()
    for codec, t in ((ol.LZ4, 0), (ol.SNAPPY, 4), (ol.ZSTD, 6)):                 # aocl_compression_type: LZ4 0, SNAPPY 4, ZSTD 6
        d = api.LlcDesc()
        cap = len(data) + len(data) // 6 + 64 + 16 + 12 * 65535
        src = C.create_string_buffer(data, len(data)); out = C.create_string_buffer(cap); back = C.create_string_buffer(len(data))
        d.inBuf = C.cast(src, C.c_void_p); d.inSize = len(data); d.outBuf = C.cast(out, C.c_void_p); d.outSize = cap
        d.level = 1 if codec == ol.ZSTD else 0
        d.numThreads = 8
        assert lib.aocl_llc_setup(C.byref(d), t) == 0
        r = lib.aocl_llc_compress(C.byref(d), t)
        assert r > 0
        stream = out.raw[:r]
        base = 8 if codec == ol.ZSTD else 0
        assert stream[base:base + 8] == b"AOCL_LLC"
        assert ref.decompress(codec, stream[ref_mt.skip(stream):], len(data)) == data
        d.inBuf = C.cast(out, C.c_void_p); d.inSize = r; d.outBuf = C.cast(back, C.c_void_p); d.outSize = len(data)
        assert lib.aocl_llc_decompress(C.byref(d), t) == len(data) and back.raw == data
        # numThreads <= 1: the one-shot stream, byte-identical to the single-threaded reference (small input)
        small = data[:50000]
        s2 = C.create_string_buffer(small, len(small))
        d.inBuf = C.cast(s2, C.c_void_p); d.inSize = len(small); d.outBuf = C.cast(out, C.c_void_p); d.outSize = cap; d.numThreads = 1
        r = lib.aocl_llc_compress(C.byref(d), t)
        assert out.raw[:r] == ref.compress(codec, small)
        lib.aocl_llc_destroy(C.byref(d), t)
