"""bench.py's own launcher (verdict r04 item 1): `python3 bench.py --gpus N` without WORLD_SIZE / RANK in the environment
must start the ranks itself as CHILD processes - the parent never imports torch (so it never initialises the GPU) - and
relay the children's exit code.  CPU-only: the dry run prints the command; the real run's ranks fail (no GPU here) and the
parent must come back non-zero instead of hanging or printing a number."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = dict(os.environ, **kw)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_dry_run_builds_the_launcher_command_without_importing_torch():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1"],
                       env=_env(AOCLGPU_BENCH_LAUNCH_DRYRUN="1", WORLD_SIZE="1"), capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    rec = json.loads(p.stdout.strip().splitlines()[-1])
    cmd = rec["launch"]
    assert rec["torch_imported_by_parent"] is False
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert "--standalone" in cmd and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1"      # the launcher picks (and keeps) its own port
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"]


def test_under_a_foreign_launcher_it_refuses_to_fan_out():
    # mpirun / srun set their own rank variables, not RANK: every one of their processes would start N ranks of its own
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"],
                       env=_env(AOCLGPU_BENCH_LAUNCH_DRYRUN="1", OMPI_COMM_WORLD_RANK="1"), capture_output=True, text=True, timeout=120)
    assert p.returncode == 2 and '"launch"' not in p.stdout and "OMPI_COMM_WORLD_RANK" in p.stdout


def test_under_a_launcher_it_does_not_launch_again():
    # RANK present = a launcher started us: no second launch (the dry-run switch would print the command otherwise); the rank
    # itself then fails on this GPU-less box, which is the loud failure the product path owes
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline"],
                       env=dict(_env(AOCLGPU_BENCH_LAUNCH_DRYRUN="1"), RANK="0", WORLD_SIZE="1"), capture_output=True, text=True, timeout=300)
    assert '"launch"' not in p.stdout
    assert p.returncode != 0


def test_exit_code_of_failed_ranks_is_relayed():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--gib", "0.01", "--base-mib", "1"],
                       env=_env(AOCLGPU_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
