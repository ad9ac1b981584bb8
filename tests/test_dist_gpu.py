"""BASELINE config 5 as far as one GPU allows: two ranks (gloo rendezvous, both on GPU 0) run bench.py's N>1 step with the
REAL kernels - compress, aoclgpu_compact_slots, exchange of the compacted segments, decompress - and bench.py's own checks
outside the timed region must have passed: every rank decodes the segment it RECEIVED from the other rank and compares it
with that rank's source, and one RAP stream assembled across the rank boundary (aoclgpu_rap_assemble) decodes to both
ranks' bytes.  A wrong exchange cannot produce the JSON line."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("exchange", ["sendrecv", "allgather"])
def test_two_ranks_one_gpu(exchange):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, AOCLGPU_BENCH_BACKEND="gloo", AOCLGPU_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--gib", "0.0625", "--base-mib", "24", "--exchange", exchange]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and set(out["codecs"]) == {"lz4", "zstd"}
    for name, rec in out["codecs"].items():
        assert "exchange" in rec["checks"] and "rap_assemble" in rec["checks"] and "round_trip" in rec["checks"], (name, rec["checks"])
        assert rec["value"] > 0
