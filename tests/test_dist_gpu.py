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


def test_bench_starts_its_own_ranks():
    """`python3 bench.py --gpus 2` with NO launcher and no WORLD_SIZE: bench.py starts the rank processes itself (a child
    torch.distributed.run; the parent makes no GPU call) and relays rank 0's line; the N>1 line is self-describing."""
    env = dict(os.environ, AOCLGPU_BENCH_BACKEND="gloo", AOCLGPU_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--gib", "0.0625", "--base-mib", "24"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["launched_by"].startswith("bench.py itself")
    for key in ("rccl_nranks", "exchange_ms", "exchange_GBps", "exchange_exposed_ms", "exchange_fell_back", "exchange_fallback_reason", "exchange"):
        assert key in out, key
        for rec in out["codecs"].values():
            assert key in rec, key
    assert out["exchange_ms"] > 0 and out["exchange_GBps"] > 0
    # gloo validation run: the C exchange was requested and could not be used - the line must say so
    assert out["exchange_requested"] == "c" and out["exchange_fell_back"] is True and "gloo" in out["exchange_fallback_reason"]
    assert out["rccl_nranks"] is None


def test_n1_line_has_no_exchange_keys():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--gib", "0.0625",
                        "--base-mib", "24", "--codec", "lz4", "--no-cpu-baseline", "--no-by-kind"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1 and "exchange_ms" not in out and "launched_by" not in out
