"""CPU: the Python restatement of the RAP join (tests/rap_model.py) against the real threaded reference build."""
import pytest

import aocl_compression_amd as a
import oracle_lib as ol
import rap_model


@pytest.mark.parametrize("kind", ["text", "binary", "random"])
@pytest.mark.parametrize("codec", [ol.LZ4, ol.SNAPPY])
def test_model_equals_threaded_reference(oracle, ref, ref_mt, codec, kind):
    T, part = 3, 300000
    data = a.corpus.make(kind, T * part, seed=21).tobytes()
    ref_mt.set_threads(T)
    want = ref_mt.compress(codec, data, len(data) + len(data) // 6 + 4096)
    assert rap_model.pack(oracle, codec, data, part) == want


def test_model_streams_decode_with_the_reference(oracle, ref, ref_mt):
    data = a.corpus.silesia_mix(700001, seed=4, segment=30000).tobytes()
    for codec in (ol.LZ4, ol.SNAPPY, ol.ZSTD):
        for chunk in (65536, 1000, 131072):
            s = rap_model.pack(oracle, codec, data, chunk)
            assert ref.decompress(codec, s[ref_mt.skip(s):], len(data)) == data
