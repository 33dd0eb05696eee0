"""`--gpus N` outside torchrun: one fresh process per GPU, started BEFORE the parent has made any GPU call.

Imports nothing of the package (the parent must stay free of the HIP runtime: it only starts, watches and reaps its ranks), so
the command-line drivers load this file by path (`load_launcher()` in bench.py / depth_to_reconstruction.py).

What the reference does here: nothing -- it is single-process.  Rendezvous on 127.0.0.1.
"""
import os
import socket
import subprocess
import sys
import time


def free_port_socket():
    """A bound socket on 127.0.0.1 with a kernel-chosen port.  The CALLER keeps it open until the ranks have been started (bind ->
    close -> reuse is a race with every other process that asks for a free port; with SO_REUSEADDR on both sides rank 0's listener
    can bind while this one is still held)."""
    sk = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    sk.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    sk.bind(("127.0.0.1", 0))
    return sk, sk.getsockname()[1]


def exit_code(returncodes):
    """Exit status of a group of ranks: 0 only if every rank returned 0.  A rank killed by a signal has a NEGATIVE return code
    (a GPU fault ends in SIGSEGV / SIGABRT): max() over the group would report the survivors' 0."""
    for c in returncodes:
        if c:
            return abs(int(c)) or 1
    return 0


def spawn_ranks(script, argv, n, timeout_s=None, extra_env=None, poll_s=0.05, grace_s=5.0):
    """Start `n` copies of `python script argv...` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), wait for all of
    them and return the group's exit status.  The first rank that ends with a non-zero status (negative included) ends the group:
    its siblings -- which would otherwise sit in a collective for ever -- are terminated, then killed.  `timeout_s` bounds the
    whole wait (None: TL3D_RANK_TIMEOUT_S from the environment, default 3600 s)."""
    if timeout_s is None:
        timeout_s = float(os.environ.get("TL3D_RANK_TIMEOUT_S", "3600"))
    sk, port = free_port_socket()
    procs = []
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            if extra_env:
                env.update(extra_env)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env))
    finally:
        sk.close()                                      # the children exist; rank 0 binds the port when it reaches the rendezvous
    deadline = time.monotonic() + timeout_s
    codes = [None] * n
    failed = None
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
                if codes[i] not in (None, 0) and failed is None:
                    failed = i
        if failed is not None or time.monotonic() > deadline:
            break
        time.sleep(poll_s)
    if any(c is None for c in codes):                   # a rank failed or the group ran out of time: end the others
        why = f"rank {failed} ended with status {codes[failed]}" if failed is not None else f"no result after {timeout_s:.0f} s"
        print(f"[tl3d.launch] {why}: terminating the other ranks", file=sys.stderr)
        for i, p in enumerate(procs):
            if codes[i] is None:
                p.terminate()
        t_end = time.monotonic() + grace_s
        for i, p in enumerate(procs):
            if codes[i] is None:
                try:
                    p.wait(max(0.0, t_end - time.monotonic()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
                codes[i] = p.returncode if failed is not None else (p.returncode or 124)
        if failed is None:
            return 124
        return exit_code([codes[failed]])               # (the siblings' -SIGTERM is this function's doing, not a result)
    return exit_code(codes)
