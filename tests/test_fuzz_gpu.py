"""Property-based parity: inputs built from an LZ-shaped grammar (random literals, copies of earlier data at random
distances, byte runs, repeated periods) so that matches start and end at every kind of boundary the parsers special-case
(chunk start, MFLIMIT / kInputMarginBytes / ilimit at the end, table-type thresholds, 64 KiB fragment edges).
GPU output must equal the oracle's byte for byte and decode back."""
import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st, HealthCheck

import aocl_compression_amd as a
import oracle_lib as ol

pytestmark = pytest.mark.gpu
CODECS = {"lz4": (a.LZ4, ol.LZ4), "snappy": (a.SNAPPY, ol.SNAPPY), "zstd": (a.ZSTD, ol.ZSTD)}

op = st.one_of(
    st.tuples(st.just("lit"), st.integers(1, 40), st.integers(0, 2 ** 32 - 1)),
    st.tuples(st.just("copy"), st.integers(1, 70000), st.integers(3, 300)),
    st.tuples(st.just("run"), st.integers(0, 255), st.integers(1, 500)),
    st.tuples(st.just("period"), st.integers(1, 9), st.integers(4, 200)),
)


def build(ops, alphabet):
    out = bytearray()
    for o in ops:
        if o[0] == "lit":
            rng = np.random.default_rng(o[2])
            out += bytes(rng.integers(0, alphabet, o[1], dtype=np.uint8))
        elif o[0] == "copy" and out:
            d = 1 + (o[1] - 1) % len(out)
            for _ in range(o[2]):
                out.append(out[-d])
        elif o[0] == "run":
            out += bytes([o[1]]) * o[2]
        elif o[0] == "period" and len(out) >= o[1]:
            for _ in range(o[2]):
                out.append(out[-o[1]])
    return bytes(out)


@pytest.mark.parametrize("codec", ["lz4", "snappy", "zstd"])
@settings(max_examples=int(__import__("os").environ.get("FUZZ_EXAMPLES", "150")), deadline=None, derandomize="FUZZ_EXAMPLES" not in __import__("os").environ,
          suppress_health_check=list(HealthCheck))
@given(ops=st.lists(op, min_size=30, max_size=700), alphabet=st.sampled_from([2, 4, 16, 256]),
       chunk=st.sampled_from([0, 97, 1000, 4096, 65536]))
def test_grammar_inputs_bit_exact(eng, oracle, codec, ops, alphabet, chunk):
    gc, oc = CODECS[codec]
    raw = build(ops, alphabet)
    if not raw:
        return
    if chunk == 0 or (codec == "zstd" and len(raw) <= 131072 and chunk == 65536):
        got = eng.compress_host(gc, raw[:131072] if codec == "zstd" else raw)
        want = oracle.compress(oc, raw[:131072] if codec == "zstd" else raw)
        assert got == want
        assert eng.decompress_host(gc, got, len(raw[:131072] if codec == "zstd" else raw)) == (raw[:131072] if codec == "zstd" else raw)
        return
    src = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(eng.device)
    slots, out_len, slot = eng.compress_chunks(gc, src, chunk)
    lens = out_len.cpu().tolist(); host = slots.cpu().numpy()
    for i in range(0, len(raw), chunk):
        k = i // chunk
        assert host[k * slot:k * slot + lens[k]].tobytes() == oracle.compress(oc, raw[i:i + chunk]), f"chunk {k}"
    dst, dl = eng.decompress_chunks(gc, slots, out_len, slot, len(raw), chunk)
    assert torch.equal(dst, src)
