"""Build libtacorl_hip.so (gfx950) in-tree with hipcc.  `python -m tacorl_amd.build`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, "csrc", f) for f in ("dense_ops.hip", "rl_ops.hip", "seq_ops.hip", "encoder_fused.hip", "encoder_bwd_fused.hip", "mlp_fused.hip", "rnn_ops.hip", "pr_fused.hip")]
OUT = os.path.join(HERE, "lib", "libtacorl_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17"]


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc"))]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "tacorl_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    srcs = [s for s in SRC if os.path.exists(s)]
    if not force and not _stale():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [HIPCC, *FLAGS, "-o", OUT, *srcs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
