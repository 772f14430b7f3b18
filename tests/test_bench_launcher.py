"""bench.py's launcher path: a plain `python bench.py --gpus N` (no WORLD_SIZE) must itself start N rank processes, before
it imports torch or makes any GPU call, forward ONE JSON line and propagate the ranks' exit code.  No GPU here: the ranks run
bench.py's `--launch-check` (rendezvous over gloo on the CPU, report RANK / WORLD_SIZE, exit).  SURVEY.md 8(e): the reference has
no multi-GPU mode, so the N-GPU command is the build's own to carry."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, extra_env=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def json_lines(text):
    out = []
    for ln in text.splitlines():
        if ln.lstrip().startswith("{"):
            try:
                out.append(json.loads(ln))
            except ValueError:
                pass
    return out


def test_plain_invocation_starts_its_own_ranks():
    p = run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--launch-check"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = json_lines(p.stdout)
    assert len(lines) == 1 and len(p.stdout.strip().splitlines()) == 1, p.stdout  # exactly one line on stdout, the result
    r = lines[0]
    assert r["launch_check"] and r["world"] == 2 and r["ranks"] == [0, 1] and r["env_world_size"] == 2 and r["gpus_arg"] == 2
    assert r["cuda_initialized"] is False
    # the parent became the launcher before importing torch (hence before any GPU call)
    assert "[bench launcher] torch imported: False" in p.stderr


def test_launcher_propagates_a_rank_failure():
    p = run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--launch-check", "--launch-check-fail-rank", "1"])
    assert p.returncode != 0
    assert len(json_lines(p.stdout)) <= 1


def test_torchrun_form_still_works_and_a_mismatched_gpus_flag_does_not_abort():
    """The driver's own N > 1 command (torch.distributed.run around bench.py); --gpus disagreeing with WORLD_SIZE is reported,
    not asserted on."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", "29613", BENCH, "--gpus", "4", "--launch-check"],
                       env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json_lines(p.stdout)
    assert len(r) == 1 and r[0]["world"] == 2 and r[0]["gpus_arg"] == 4


def test_launcher_never_imports_torch_in_the_parent():
    """Static side of the same guarantee: bench.py's module level imports nothing but the standard library."""
    import ast
    tree = ast.parse(open(BENCH).read())
    top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
    names = {a.name.split(".")[0] for n in top if isinstance(n, ast.Import) for a in n.names} | {
        n.module.split(".")[0] for n in top if isinstance(n, ast.ImportFrom)}
    assert names <= {"argparse", "json", "os", "socket", "subprocess", "sys", "time"}, names


@pytest.mark.parametrize("name,mflop", [("hopper", 4.11), ("can", 4.44), ("halfcheetah", 4.21)])
def test_flop_counts_follow_survey_8d(name, mflop):
    sys.path.insert(0, ROOT)
    import bench
    assert abs(bench.flop_per_sample(bench.WORKLOADS[name]) / 1e6 - mflop) < 0.01
    if name == "hopper":
        assert abs(bench.flop_per_chunk(bench.WORKLOADS[name]) / 1e6 - 22.06) < 0.01
