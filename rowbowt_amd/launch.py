"""One process per GPU, started from a parent that never touches the GPU.

`python bench.py --gpus N` (and tools/pangenome_stream.py) call `run_ranks()` when N > 1 and no RANK is set:
the parent -- which has imported neither torch nor librbg.so -- starts N fresh children of the same
script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, relays rank 0's stdout (the one
JSON line), and exits non-zero when any child does.  A process that has initialised HIP is never re-exec'd
(that takes a GPU box down); the children are ordinary `subprocess.Popen` children.

Under `python -m torch.distributed.run` the ranks already exist: `check_world()` only verifies that
`--gpus` equals WORLD_SIZE.  This module is stdlib-only on purpose and is loaded by file path, so that using it
cannot import the package's ctypes binding.

The read loop these ranks shard is the reference's `rb_align.cpp:176-178`.
"""
import json
import os
import socket
import subprocess
import sys
import threading

RANK_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")


class LaunchError(SystemExit):
    """raised (as a non-zero exit) when the requested world cannot be formed"""

    def __init__(self, msg, code=2):
        print(f"[launch] {msg}", file=sys.stderr, flush=True)
        super().__init__(code)


def under_launcher(env=None):
    env = os.environ if env is None else env
    return "RANK" in env


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpus(python=None, timeout=600):
    """Number of GPUs a FRESH child sees (torch.cuda.device_count() in a throw-away process: the parent
    stays free of HIP).  None when it cannot be determined."""
    code = "import torch; print(torch.cuda.device_count())"
    try:
        out = subprocess.run([python or sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout)
        return int(out.stdout.strip().splitlines()[-1]) if out.returncode == 0 else None
    except Exception:
        return None


def check_world(gpus, env=None):
    """For a process that already is a rank (RANK set): `--gpus` must be the world it was started in."""
    env = os.environ if env is None else env
    world = int(env.get("WORLD_SIZE", "1"))
    if gpus != world:
        raise LaunchError(f"--gpus {gpus} but WORLD_SIZE={world}: start {gpus} ranks (python bench.py --gpus {gpus} does it "
                          f"itself) or pass --gpus {world}")
    return int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", env.get("RANK", "0"))), world


def rank_env(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this pool
    return env


def _relay(src, dst, prefix=None):
    for line in iter(src.readline, ""):
        dst.write(line if prefix is None else f"{prefix}{line}")
        dst.flush()
    src.close()


def run_ranks(script, argv, gpus, check_devices=True, timeout=None, popen=subprocess.Popen):
    """Start `gpus` children `python script *argv`, one rank each; relay rank 0's stdout to ours, every other
    rank's stdout and all stderr to our stderr; return the worst exit code.  A failing rank takes the others
    down (killed by PID: they would wait in a collective for ever)."""
    if gpus < 1:
        raise LaunchError(f"--gpus {gpus}: need at least one")
    assert "torch" not in sys.modules, "the launching parent must not have imported torch"
    if check_devices:
        have = visible_gpus()
        if have is not None and have < gpus:
            raise LaunchError(f"--gpus {gpus} but this node shows {have} GPU(s): refusing to run a mislabelled {have}-GPU job")
    port = free_port()
    procs, threads = [], []
    for r in range(gpus):
        p = popen([sys.executable, script, *argv], env=rank_env(r, gpus, port), stdout=subprocess.PIPE, stderr=None, text=True, bufsize=1)
        procs.append(p)
        t = threading.Thread(target=_relay, args=(p.stdout, sys.stdout if r == 0 else sys.stderr, None if r == 0 else f"[rank {r}] "), daemon=True)
        t.start()
        threads.append(t)
    worst = 0
    pending = set(range(gpus))
    import time
    t0 = time.time()
    while pending:
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is None:
                continue
            pending.discard(r)
            if rc != 0:
                worst = worst or rc
                print(f"[launch] rank {r} exited with {rc}; stopping the other ranks", file=sys.stderr, flush=True)
                for q in sorted(pending):
                    procs[q].kill()
        if timeout is not None and time.time() - t0 > timeout and pending:
            print(f"[launch] timeout after {timeout}s; stopping ranks {sorted(pending)}", file=sys.stderr, flush=True)
            for q in sorted(pending):
                procs[q].kill()
            worst = worst or 124
        time.sleep(0.05)
    for t in threads:
        t.join(timeout=5)
    return worst


def echo_rank():
    """--launch-check: what a rank was started with, as one JSON line (no GPU, no torch)"""
    print(json.dumps({"launch_check": True, **{k: os.environ.get(k) for k in RANK_ENV}, "pid": os.getpid(), "ppid": os.getppid(),
                      "torch_imported": "torch" in sys.modules, "argv": sys.argv[1:]}), flush=True)
