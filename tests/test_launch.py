"""The N-rank launcher of bench.py / tools/pangenome_stream.py (rowbowt_amd/launch.py), on the CPU: `--gpus N` without
RANK starts N fresh children with the right environment from a parent that has imported neither torch nor the HIP
library; `--gpus` that disagrees with WORLD_SIZE, or exceeds the node's GPUs, is refused with a non-zero exit; a
failing rank takes the job down.  The loop the ranks shard is the reference's rb_align.cpp:176-178."""
import importlib.util
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPTS = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "tools", "pangenome_stream.py")]


def clean_env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def load_launch():
    spec = importlib.util.spec_from_file_location("rbg_launch_t", os.path.join(ROOT, "rowbowt_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("script", SCRIPTS)
@pytest.mark.parametrize("n", [2, 3])
def test_gpus_n_starts_n_ranks(script, n):
    p = subprocess.run([sys.executable, script, "--gpus", str(n), "--launch-check", "--seed", "7"], env=clean_env(),
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    out_lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    err_lines = [json.loads(l.split("] ", 1)[1]) for l in p.stderr.splitlines() if l.startswith("[rank ")]
    assert len(out_lines) == 1 and out_lines[0]["RANK"] == "0"          # only rank 0's line reaches stdout
    ranks = out_lines + err_lines
    assert sorted(r["RANK"] for r in ranks) == [str(i) for i in range(n)]
    assert all(r["LOCAL_RANK"] == r["RANK"] and r["WORLD_SIZE"] == str(n) and r["MASTER_ADDR"] == "127.0.0.1" for r in ranks)
    assert len({r["MASTER_PORT"] for r in ranks}) == 1 and len({r["ppid"] for r in ranks}) == 1 and len({r["pid"] for r in ranks}) == n
    assert not any(r["torch_imported"] for r in ranks)                  # the check path touches neither torch nor a GPU
    assert all("--seed" in r["argv"] and "7" in r["argv"] for r in ranks)   # the command line travels unchanged


@pytest.mark.parametrize("script", SCRIPTS)
def test_gpus_must_equal_world_size(script):
    # as torch.distributed.run would start it, but with the wrong --gpus
    p = subprocess.run([sys.executable, script, "--gpus", "2", "--launch-check"], capture_output=True, text=True, timeout=60,
                       env=clean_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533"))
    assert p.returncode != 0 and "WORLD_SIZE=4" in p.stderr
    p = subprocess.run([sys.executable, script, "--gpus", "4", "--launch-check"], capture_output=True, text=True, timeout=60,
                       env=clean_env(RANK="3", LOCAL_RANK="3", WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533"))
    assert p.returncode == 0 and json.loads(p.stdout)["RANK"] == "3"


def test_more_gpus_than_the_node_has_is_refused_by_the_parent():
    """here: no GPU at all.  The parent asks a throw-away child for torch.cuda.device_count() and refuses; it never
    imports torch itself (run_ranks asserts that)."""
    p = subprocess.run([sys.executable, SCRIPTS[0], "--gpus", "2"], env=clean_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and p.stdout.strip() == ""
    assert "GPU(s)" in p.stderr and "refusing" in p.stderr


def test_failing_rank_stops_the_job(tmp_path):
    launch = load_launch()
    child = tmp_path / "child.py"
    child.write_text("import os, sys, time\n"
                     "r = int(os.environ['RANK'])\n"
                     "print('hello from', r, flush=True)\n"
                     "if r == 1:\n    sys.exit(3)\n"
                     "time.sleep(120)\n")
    # in a fresh interpreter: run_ranks refuses to run in a process that has imported torch (this pytest process may have)
    driver = ("import importlib.util, sys\n"
              f"spec = importlib.util.spec_from_file_location('l', {os.path.join(ROOT, 'rowbowt_amd', 'launch.py')!r})\n"
              "m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)\n"
              f"sys.exit(m.run_ranks({str(child)!r}, [], 3, check_devices=False))\n")
    t0 = time.time()
    p = subprocess.run([sys.executable, "-c", driver], env=clean_env(), capture_output=True, text=True, timeout=100)
    assert p.returncode == 3 and time.time() - t0 < 60
    assert "hello from 0" in p.stdout and "[rank 1] hello from 1" in p.stderr and "rank 1 exited with 3" in p.stderr
    import torch  # noqa: F401  -- and with torch imported the parent refuses to launch at all
    with pytest.raises(AssertionError):
        launch.run_ranks(str(child), [], 2, check_devices=False)


def test_rank_env_and_check_world():
    launch = load_launch()
    e = launch.rank_env(2, 8, 1234, base={})
    assert (e["RANK"], e["LOCAL_RANK"], e["WORLD_SIZE"], e["MASTER_ADDR"], e["MASTER_PORT"]) == ("2", "2", "8", "127.0.0.1", "1234")
    assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert launch.check_world(8, env=e) == (2, 2, 8)
    with pytest.raises(SystemExit):
        launch.check_world(4, env=e)
    with pytest.raises(SystemExit):
        launch.run_ranks("x.py", [], 0, check_devices=False)


@pytest.mark.gpu
def test_on_the_gpu_box_one_rank_more_than_devices_is_refused_and_launch_check_names_every_rank():
    """the same refusal where GPUs exist: --gpus (devices + 1) never starts a rank; --gpus devices passes the check and the
    children are given ranks 0 .. devices - 1 (--launch-check: no GPU is touched by anyone)."""
    count = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=300)
    ndev = int(count.stdout.strip())
    assert ndev >= 1
    p = subprocess.run([sys.executable, SCRIPTS[0], "--gpus", str(ndev + 1)], env=clean_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and p.stdout.strip() == "" and "refusing" in p.stderr
    if ndev >= 2:
        p = subprocess.run([sys.executable, SCRIPTS[0], "--gpus", str(ndev), "--launch-check"], env=clean_env(), capture_output=True, text=True, timeout=600)
        assert p.returncode == 0
        ranks = sorted(json.loads(line.split("] ", 1)[-1])["RANK"] for line in (p.stdout + p.stderr).splitlines() if '"RANK"' in line)
        assert ranks == sorted(str(r) for r in range(ndev))


def test_replicas_and_gpus_are_two_ways_and_exclude_each_other():
    """tools/pangenome_stream.py: `--replicas G` (ONE process: one index build, peer copies, a thread per replica) and `--gpus N`
    (one process per GPU) both shard the reads of rb_align.cpp:176-178 over replicas; asking for both is refused before anything is
    imported or started."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pangenome_stream.py"), "--replicas", "2", "--gpus", "2", "--launch-check"],
                       env=clean_env(), capture_output=True, text=True, timeout=60)
    assert p.returncode != 0 and "--replicas" in p.stderr and "--gpus" in p.stderr and p.stdout.strip() == ""


def test_a_rank_that_fails_names_its_stage_before_the_parent_stops_the_others():
    """VERDICT r5 item 5 (the first 8-GPU run is the driver's): a rank of bench.py that dies says WHICH rank, WHERE in its run (bench.py `stage()`) and
    why on stderr; the launching parent then reports the exit code and stops the other ranks.  Here: two rehearsal ranks on a box without a GPU --
    both fail at the stage that asks for the device."""
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU box: the ranks would run")
    p = subprocess.run([sys.executable, SCRIPTS[0], "--gpus", "2", "--rehearse-ranks", "--L", "20000", "--H", "3", "--reads", "100"],
                       capture_output=True, text=True, timeout=600, env=clean_env())
    assert p.returncode != 0
    assert "FAILED at stage 'import torch'" in p.stderr and "bench.py needs an MI355X" in p.stderr
    assert "[bench] rank 0 (pid" in p.stderr or "[bench] rank 1 (pid" in p.stderr
    assert "[launch] rank" in p.stderr and "stopping the other ranks" in p.stderr


def test_replicas_flag_is_refused_under_a_multi_rank_world():
    """bench.py --replicas G (one process, G replicas) with --gpus N > 1 used to be ignored silently: refused now, like pangenome_stream.py"""
    env = clean_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, SCRIPTS[0], "--gpus", "2", "--replicas", "2"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode != 0 and "pick one" in p.stderr
