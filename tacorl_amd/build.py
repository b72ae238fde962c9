"""Build libtacorl_hip.so (gfx950) in-tree with hipcc.  `python -m tacorl_amd.build [--force]`.
Each csrc/*.hip is compiled to its own object (only when it or a header changed, up to 4 at a time), then linked."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
NAMES = ("dense_ops", "rl_ops", "seq_ops", "encoder_fused", "encoder_ring", "encoder_bwd_fused", "mlp_fused", "rnn_ops",
         "pr_fused", "data_ops")
SRC = [os.path.join(CSRC, f + ".hip") for f in NAMES]
OBJ_DIR = os.path.join(HERE, "lib", "obj")
OUT = os.path.join(HERE, "lib", "libtacorl_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17"]
# per-file extras.  encoder_fused: neither the IR load/store vectorizer nor the SI load/store optimizer - either one
# fuses the conv1 fragment reads (8-byte halves at a 24-byte pixel stride) into ds_read2_b64, which the LDS serves at
# 128 B/clk in 16-lane groups; plain ds_read_b64 pairs run at 256 B/clk in the 32-lane groups the tile layout is made for.
_NO_LSOPT = ["-Xclang", "-target-feature", "-Xclang", "-load-store-opt", "-mllvm", "-amdgpu-load-store-vectorizer=0"]
EXTRA = {"encoder_fused": _NO_LSOPT, "encoder_ring": _NO_LSOPT}


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "tacorl_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def build(force=False, verbose=True):
    srcs = [s for s in SRC if os.path.exists(s)]
    os.makedirs(OBJ_DIR, exist_ok=True)
    hm = _headers_mtime()
    todo = []
    objs = []
    for s in srcs:
        o = os.path.join(OBJ_DIR, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hm):
            todo.append((s, o))

    def cc(so):
        cmd = [HIPCC, *FLAGS, *EXTRA.get(os.path.basename(so[0])[:-4], []), "-c", so[0], "-o", so[1]]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    if todo:
        with ThreadPoolExecutor(max_workers=4) as ex:
            list(ex.map(cc, todo))
    if todo or not os.path.exists(OUT) or any(os.path.getmtime(o) > os.path.getmtime(OUT) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
