"""`python bench.py --gpus N` starts its own ranks (the reference is launched once and fans out:
train.py:25-29, README.md:36-42).  CPU: the launcher starts N fresh children with the torchrun environment, lets their stdout
through and reports failure; GPU: the whole bench at world size 2 over gloo on the one MI355X prints ONE line with n_gpus 2."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _script(tmp_path, body):
    p = tmp_path / "child.py"
    p.write_text(textwrap.dedent(body))
    return str(p)


def test_self_launch_starts_n_ranks_and_relays_rank0(tmp_path, capfd):
    sys.path.insert(0, ROOT)
    import bench
    script = _script(tmp_path, """
        import os, sys, json
        r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["MASTER_ADDR"] == "127.0.0.1" and os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        open(os.path.join(sys.argv[1], f"rank{r}"), "w").write(str(w))
        if r == 0:
            print(json.dumps({"n_gpus": w, "args": sys.argv[2:]}), flush=True)
    """)
    rc = bench.self_launch(2, [str(tmp_path), "--gpus", "2"], script=script)
    assert rc == 0
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("rank")) == ["rank0", "rank1"]
    out = capfd.readouterr().out
    line = [l for l in out.splitlines() if l.startswith("{")]
    assert len(line) == 1 and json.loads(line[0]) == {"n_gpus": 2, "args": ["--gpus", "2"]}


def test_self_launch_reports_a_failed_rank(tmp_path):
    sys.path.insert(0, ROOT)
    import bench
    script = _script(tmp_path, """
        import os, sys
        sys.exit(3 if os.environ["RANK"] == "1" else 0)
    """)
    assert bench.self_launch(2, [], script=script) != 0


def test_main_refuses_a_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr


@pytest.mark.gpu
def test_bench_gpus_2_over_gloo_prints_one_line(dev):
    env = dict(os.environ, PT_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--workload", "A", "--no-decode", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["nccl_ranks"] == 2 and out["config"]["parallelism"] == "dp2"
    assert out["allreduce_ms"] > 0 and 0.0 <= out["allreduce_overlap_pct"] <= 100.0
    assert out["config"]["global_batch"] == 2 * out["config"]["per_gpu_batch"]
