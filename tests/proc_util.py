"""Child-process launches of the multi-rank tests: the launcher and its rank processes run in their own process group,
which is killed as a whole on a timeout and after every run - a rank left behind by a failed launch would keep the GPU
(and the rendezvous port) and hang every later test of the session."""
import os
import signal
import socket
import subprocess


def free_port():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


class Result:
    def __init__(self, returncode, stdout, stderr):
        self.returncode, self.stdout, self.stderr = returncode, stdout, stderr


def run_group(cmd, env, cwd, timeout):
    p = subprocess.Popen(cmd, env=env, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
        rc = p.returncode
    except subprocess.TimeoutExpired:
        _kill(p.pid)
        out, err = p.communicate()
        rc, err = -9, (err or "") + f"\n[timeout after {timeout}s: process group killed]"
    finally:
        _kill(p.pid)
    return Result(rc, out or "", err or "")


def _kill(pgid):
    try:
        os.killpg(pgid, signal.SIGKILL)
    except (ProcessLookupError, PermissionError):
        pass
