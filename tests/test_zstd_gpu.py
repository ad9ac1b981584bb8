"""GPU parity: HIP Zstd level-1 kernels (through the C-ABI) vs the CPU oracle -- bit-exact frames."""
import numpy as np
import pytest
import torch

import aocl_compression_amd as a
import oracle_lib as ol

pytestmark = pytest.mark.gpu

KNOWN = [(b"", "28b52ffd2000010000"), (b"a" * 100, "28b52ffd206445000010616101003f012c"),
         (b"abcd" * 8, "28b52ffd202055000020616263640100338e08")]   # SURVEY.md 8c, from the compiled reference


def _chunks(raw, chunk):
    return [raw[i:i + chunk] for i in range(0, max(len(raw), 1), chunk)]


def test_known_answers(eng):
    for inp, hexout in KNOWN:
        assert eng.compress_host(a.ZSTD, inp).hex() == hexout


@pytest.mark.parametrize("kind", list(a.corpus.KINDS))
@pytest.mark.parametrize("chunk", [131072, 65536, 16384, 4096])
def test_compress_bit_exact(eng, oracle, kind, chunk):
    data = a.corpus.make(kind, (1 << 20) + 555, seed=13)
    src = torch.from_numpy(data).to(eng.device)
    slots, out_len, slot = eng.compress_chunks(a.ZSTD, src, chunk)
    lens = out_len.cpu().tolist()
    host = slots.cpu().numpy()
    parts = _chunks(data.tobytes(), chunk)
    bad = []
    for i, part in enumerate(parts):
        want = oracle.compress(ol.ZSTD, part)
        got = host[i * slot:i * slot + lens[i]].tobytes()
        if got != want:
            bad.append((i, len(part), lens[i], len(want)))
    assert not bad, bad[:5]


@pytest.mark.parametrize("size", [1, 6, 7, 19, 20, 21, 63, 64, 65, 255, 256, 257, 1000, 1023, 1024, 4095, 16383, 16384,
                                  16385, 65535, 65791, 65792, 100001, 131071, 131072])
def test_ragged_sizes(eng, oracle, size):
    raw = a.corpus.make("text", size, seed=size).tobytes()
    assert eng.compress_host(a.ZSTD, raw) == oracle.compress(ol.ZSTD, raw)


def test_special_inputs(eng, oracle):
    rng = np.random.default_rng(5)
    cases = [
        bytes(131072), b"\xff" * 70000, bytes(range(256)) * 512, b"ab" * 50000, b"abc" * 40000,
        rng.integers(0, 256, 131072, dtype=np.uint8).tobytes(),
        rng.integers(0, 2, 131072, dtype=np.uint8).tobytes(),
        rng.integers(0, 16, 131072, dtype=np.uint8).tobytes(),
        (b"x" * 100 + rng.integers(0, 256, 3000, dtype=np.uint8).tobytes()) * 40,
        bytes(70000) + rng.integers(0, 256, 60000, dtype=np.uint8).tobytes(),
        rng.integers(0, 256, 66000, dtype=np.uint8).tobytes() + bytes(65000),
        rng.integers(0, 256, 40942, dtype=np.uint8).tobytes() + a.corpus.make("text", 40942, seed=8).tobytes(),
    ]
    for i, raw in enumerate(cases):
        assert eng.compress_host(a.ZSTD, raw) == oracle.compress(ol.ZSTD, raw), i


def test_small_output_buffer_fails(eng):
    raw = a.corpus.make("text", 5000, seed=3).tobytes()
    assert eng.compress_host(a.ZSTD, raw, a.compress_bound(a.ZSTD, len(raw)) - 1) is None
