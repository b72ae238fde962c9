"""RCCL on the step's path with ONE rank (the pool's boxes have one GPU): torch.distributed backend `nccl` (= RCCL on ROCm),
world_size 1, TACORL_FORCE_COLLECTIVES=1 so that every collective an N-GPU step issues is really issued - the parameter
broadcast at construction, all-reduce #1 (d log alpha) and #2 (the gradient arena) between the three hipGraph segments of
a step, PlayLMP's arena all-reduce, the log vector's reduction on logging steps.  A sum over one rank is the identity, so
three graph-mode steps must end where the collective-free single-graph steps end.  Reference: `strategy: ddp`
(config/trainer/default.yaml:1-3, scripts/train.py:75), modules/tacorl/tacorl.py:196-202 (sync_dist).

    python tests/rccl_one_rank_script.py             eager all-reduces between graph segments (the N-GPU default)
    python tests/rccl_one_rank_script.py --in-graph  the all-reduces captured as nodes of the step's one graph

Launched by tests/test_dist_gpu.py; prints one `CASE <name> <form>: ok ...` line per case and `ALL OK`."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests.dist_shard_script import build, rel  # noqa: E402
from tests.golden_util import Golden  # noqa: E402
from tests.test_step_gpu import to_dev  # noqa: E402


def rccl_mapped():
    with open("/proc/self/maps") as f:
        return sorted({ln.split()[-1] for ln in f if "librccl" in ln or "libnccl" in ln})


def steps(kind, mod, g, n_steps=3):
    if "epoch" in g.cfg:
        mod.current_epoch = g.cfg["epoch"]
    mod.enable_graph()
    logs = []
    for i in range(n_steps):  # first call: eager warm-up + capture; then replays
        b, nz = g.batch(i % 2), g.noise(i % 2)
        if kind == "playlmp":
            nz = {k: nz[k] for k in ("eps_plan", "u_plan") if k in nz}
        mod.training_step(to_dev(b, mod.device), noise=to_dev(nz, mod.device))
        logs.append(dict(mod.logged))
    torch.cuda.synchronize()
    grads = {k: v.detach().clone() for k, v in mod.named_gradients().items()}
    params = {k: v.detach().clone() for k, v in mod.state_dict().items() if v.dtype.is_floating_point}
    n_graphs = [len(v[0]) + (v[1] is not None) for v in mod._graphs.values()]
    return grads, params, logs, n_graphs


def main():
    in_graph = "--in-graph" in sys.argv
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    assert dist.get_backend() == "nccl"
    t = torch.ones(8, device="cuda")
    dist.all_reduce(t)  # communicator init
    torch.cuda.synchronize()
    libs = rccl_mapped()
    assert libs, "librccl is not mapped into the process after an nccl all-reduce"
    print("rccl:", libs, flush=True)
    form = "in-graph" if in_graph else "segments"
    cases = [("tacorl", "tacorl_q"), ("tacorl", "tacorl_q_ad"), ("cql", "cql_q"), ("playlmp", "playlmp")]
    for kind, name in cases:
        g = Golden(name)
        params0 = g.params()
        os.environ["TACORL_FORCE_COLLECTIVES"] = "0"
        os.environ["TACORL_GRAPH_COLLECTIVES"] = "0"
        ref = build(kind, g, 1)
        ref.load_state_dict(params0, strict=False)
        g_ref, p_ref, l_ref, ng_ref = steps(kind, ref, g)
        assert ng_ref and all(n == 1 for n in ng_ref), f"{name}: the collective-free step should be one graph, got {ng_ref}"
        del ref
        os.environ["TACORL_FORCE_COLLECTIVES"] = "1"
        os.environ["TACORL_GRAPH_COLLECTIVES"] = "1" if in_graph else "0"
        calls = {"n": 0}
        orig = dist.all_reduce

        def counted(*a, **k):
            calls["n"] += 1
            return orig(*a, **k)

        dist.all_reduce = counted
        try:
            mod = build(kind, g, 1)
            mod.load_state_dict(params0, strict=False)
            g_c, p_c, l_c, ng_c = steps(kind, mod, g)
        finally:
            dist.all_reduce = orig
        want_graphs = 1 if in_graph else (2 if kind == "playlmp" else 3)
        assert ng_c and all(n >= want_graphs for n in ng_c), f"{name} {form}: graphs per step {ng_c}, expected >= {want_graphs}"
        if in_graph:
            assert all(n == 1 for n in ng_c), f"{name}: in-graph form must stay one graph, got {ng_c}"
        per_step = 1 if kind == "playlmp" else 2
        # eager form: every step issues its collectives from python (+1 per logging step for the log vector); in-graph: only the
        # warm-up and the capture do, the replays run them as graph nodes
        min_calls = per_step * (2 if in_graph else 3)
        assert calls["n"] >= min_calls, f"{name} {form}: {calls['n']} all-reduce calls, expected >= {min_calls}"
        bad = []
        for k, v in g_ref.items():
            if v.norm() > 0 and rel(g_c[k], v) > 1e-6:
                bad.append(f"grad {k}: rel {rel(g_c[k], v):.3g}")
        worst = 0.0
        for k, v in p_ref.items():
            r = rel(p_c[k], v)
            worst = max(worst, r)
            if r > 1e-6:
                bad.append(f"param {k}: rel {r:.3g}")
        for a, b in zip(l_ref, l_c):
            for k, v in a.items():
                if abs(b[k] - v) > 1e-6 * max(abs(v), 1e-3):
                    bad.append(f"log {k}: {b[k]!r} vs {v!r}")
        assert not bad, f"{name} {form}: collective step != collective-free step\n" + "\n".join(bad[:20])
        del mod
        torch.cuda.empty_cache()
        print(f"CASE {name} {form}: ok (graphs/step {ng_c}, all_reduce calls {calls['n']}, worst param rel {worst:.2g})", flush=True)
    print("ALL OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
