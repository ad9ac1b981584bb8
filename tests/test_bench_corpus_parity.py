"""Byte-level parity on the BENCH corpora (not only sizes): the head of the synthetic Silesia mix (configs 2 and 4) and
of the JSON/log corpus (config 3), compressed on the GPU with the BASELINE chunk sizes, must equal the oracle's output
chunk by chunk, byte for byte (the oracle runs the same chunks with OpenMP on the host)."""
import ctypes as C

import numpy as np
import pytest
import torch

import aocl_compression_amd as a
import oracle_lib as ol

pytestmark = pytest.mark.gpu
CASES = [("lz4", a.LZ4, ol.LZ4, 262144, "mix"), ("snappy", a.SNAPPY, ol.SNAPPY, 65536, "jsonlog"), ("zstd", a.ZSTD, ol.ZSTD, 131072, "mix")]
_CORPUS = {}


def _corpus(kind, n):
    if kind not in _CORPUS:
        _CORPUS[kind] = a.corpus.silesia_mix(n, seed=0xA0C1) if kind == "mix" else a.corpus.make("jsonlog", n, seed=0xA0C3)
    return _CORPUS[kind]


@pytest.mark.parametrize("name,gc,oc,chunk,kind", CASES)
def test_bench_corpus_bytes_equal_the_oracle(eng, oracle, name, gc, oc, chunk, kind):
    n = 96 << 20
    data = _corpus(kind, n)
    src = torch.from_numpy(data).to(eng.device)
    slots, out_len, slot = eng.compress_chunks(gc, src, chunk)
    nchunks = n // chunk
    want = np.zeros(nchunks * slot, np.uint8)
    wlen = np.zeros(nchunks, np.uint32)
    fails = oracle.L.orc_compress_chunks(oc, data.ctypes.data, n, chunk, want.ctypes.data, slot, wlen.ctypes.data, 16)
    assert fails == 0
    got_len = out_len.cpu().numpy().astype(np.uint32)
    assert np.array_equal(got_len, wlen), f"{name}: sizes differ in chunks {np.nonzero(got_len != wlen)[0][:5]}"
    got = slots.cpu().numpy()
    used = (np.arange(slot, dtype=np.uint32)[None, :] < wlen[:, None]).reshape(-1)
    bad = np.nonzero((got != want) & used)[0]
    assert bad.size == 0, f"{name}: first differing byte in chunk {bad[0] // slot} at {bad[0] % slot}"
    dst, dl = eng.decompress_chunks(gc, slots, out_len, slot, n, chunk)
    assert torch.equal(dst, src)
