#!/usr/bin/env python
"""Headline benchmark: TACO-RL offline grad steps per second (BASELINE.json metric).

Workload (BASELINE.json configs[1] / SURVEY 8d "C2"): TACORL.training_step, frozen LMP
(finetune_action_decoder=False), batch 256 per GPU, T=16 window, 84x84x3 frames, latent plan 16,
n_action_samples=4, Q phase (epoch >= bc_epochs), bf16 MFMA operands with fp32 accumulation and
fp32 master weights.  Synthetic data (U(-1,1) frames, reference initialisers, random-init LMP).

One process per GPU over RCCL.  `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the
environment starts the N rank processes itself (child processes, created before this process has made
any GPU call); under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it is one
of the ranks.  Rank 0 prints ONE JSON line.

"step" = one full training_step on one 256-sample batch per GPU (weak scaling);
value = batch-256 grad steps per second summed over GPUs (= global samples/s / 256).  For N > 1 the
line also carries a `strong` block: the same global batch of 256 split over the ranks (256/N samples
per GPU), i.e. true optimiser steps per second.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENC_FLOP_PER_IMG_84 = 13.918e6  # 2 * 6 959 104 MAC
PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3
DATA_SEED, PARAM_SEED = 1234, 0


def synth_batch(B, T, H, W, dev, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    u = lambda *s: torch.rand(*s, device=dev, generator=g) * 2 - 1  # noqa: E731
    acts = u(B, T, 7)
    acts[..., -1] = torch.where(acts[..., -1] >= 0, 1.0, -1.0)
    disp = torch.empty(B, device=dev).geometric_(0.3, generator=g).long()
    disp[torch.rand(B, device=dev, generator=g) < 0.1] = -1
    return {"states": {"rgb_static": u(B, T, 3, H, W)}, "goal": {"rgb_static": u(B, 3, H, W)}, "actions": acts,
            "disp": disp}


def build_module(dev, compute, T, world, ad_every=1, finetune=False):
    from tacorl_amd.modules.play_lmp.play_lmp_for_rl import PlayLMP
    from tacorl_amd.modules.tacorl.tacorl import TACORL

    cams = ["rgb_static"]
    actor = {"policy": {"num_layers": 3, "hidden_dim": 256}}
    critic = {"q_network": {"num_layers": 3, "hidden_dim": 256, "last_layer_activation": "Identity"}}
    pr = dict(num_heads=8, num_layers=2, encoder_hidden_size=2048, fc_hidden_size=4096, latent_plan_dim=16,
              min_std=1e-4, dropout_p=0.0, max_position_embeddings=T)
    ad = dict(n_mixtures=10, num_layers=2, hidden_size=2048, out_features=7, num_classes=10, latent_plan_dim=16,
              rnn_model="rnn_decoder", include_goal=False)
    torch.manual_seed(PARAM_SEED)
    lmp = PlayLMP(plan_proposal=actor, plan_recognition=pr, action_decoder=ad, plan_proposal_obs_modalities=cams,
                  plan_proposal_goal_modalities=cams, plan_recognition_modalities=cams, action_decoder_modalities=cams,
                  real_world=True, device=dev, compute_dtype=compute, image_dtype=compute)
    mod = TACORL(play_lmp=lmp, finetune_action_decoder=finetune, critic=critic, real_world=True, device=dev,
                 compute_dtype=compute, image_dtype=compute, world_size=world, action_loss_every_n_steps=ad_every,
                 # config/module/tacorl.yaml:8-30
                 action_decoder_lr=3e-4, actor_lr=1e-4, critic_lr=3e-4, discount=0.95, conservative_weight=1.0,
                 reward_scale=10.0, n_action_samples=4, with_lagrange=True, deterministic_backup=True, bc_epochs=5)
    mod.current_epoch = 5  # Q phase
    return mod


def time_encoder_fwd(mod, B, H, W, iters=100):
    """HIP-event timing of the step's encoder-forward launch on the stream it runs on (torch's current
    stream): ONE encoder_fused_kernel launch over all 27*B encoder images of the step (frozen LMP B*T
    frames, actor/q1/q2 over [obs;goal], actor(next) and both targets) - the roofline kernel."""
    e = mod.engine
    fused = all(e._fused_ok(c) for c in e.cams)
    fn = e.encode_fused_only if fused else e._encode_all
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(10):
        n_img = fn()
    ev0.record()
    for _ in range(iters):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / iters
    if not fused:
        n_img = sum(n for _, _, _, n in e.enc_probs) + sum(x["n"] for x in e.extra_enc)
    return ms, n_img, fused


def time_encoder_in_step(mod, batch, steps=40):
    """The same launch timed where it runs: HIP events around the encoder-forward launch INSIDE training steps (eager
    replays of the step, events on the stream the launch is issued on).  The chip holds a higher clock for a kernel
    that sits between the step's lighter phases than for 100 copies of it back to back - this is the duration the step
    pays.  An event bracket also contains the launch gaps on either side of the kernel; they are measured, not assumed:
    right behind the encoder's bracket every step brackets a CALIBRATION kernel of about the same length - one thread that
    spins on the device's constant 100 MHz clock for 120 us and records its own begin and end (tacorl_time_spin) - so
    overhead = (event bracket around the spin) - (the spin's own duration), and kernel duration = encoder bracket - overhead.
    No fitted constant (round 3 subtracted a calibrated 2.4 us from a 1-thread launch's bracket); the evidence for the
    figure is rocprofv3's kernel trace of the same command (profiles/).
    Returns (kernel ms, raw bracket ms, bracket overhead ms)."""
    from tacorl_amd import ops
    from tacorl_amd._lib import call, ptr

    e = mod.engine
    if not all(e._fused_ok(c) for c in e.cams):
        return None
    pairs, orig, was_graph = [], e._launch_fused, mod._use_graph
    marks = torch.zeros(2 * (steps + 8), dtype=torch.int64, device=mod.device)
    spin_ticks = 12000  # 120 us at 100 MHz

    def timed(c, pr, max_wg=0):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        orig(c, pr, max_wg)
        ev[1].record()
        call("tacorl_time_spin", ptr(marks), 2 * len(pairs), spin_ticks, ops.stream())
        ev[2].record()
        pairs.append(ev)

    mod._use_graph = False
    try:
        for _ in range(5):
            mod.training_step(batch)
        e._launch_fused = timed
        for _ in range(steps):
            mod.training_step(batch)
        torch.cuda.synchronize()
    finally:
        e._launch_fused, mod._use_graph = orig, was_graph
    n = len(pairs)
    raw = sum(ev[0].elapsed_time(ev[1]) for ev in pairs) / n
    spin_bracket = sum(ev[1].elapsed_time(ev[2]) for ev in pairs) / n
    m = marks[: 2 * n].cpu().view(n, 2)
    spin_actual = float((m[:, 1] - m[:, 0]).double().mean()) * 1e-5  # 100 MHz ticks -> ms
    over = min(max(spin_bracket - spin_actual, 0.0), raw)
    for _ in range(3):  # back on the captured path
        mod.training_step(batch)
    return raw - over, raw, over


def kernel_source_hash():
    import hashlib

    h = hashlib.sha256()  # the kernel's source file and the header that holds its MFMA / DPP macros and launch table
    for name in ("encoder_fused.hip", "encoder_fused.h"):
        with open(os.path.join(ROOT, "tacorl_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def measured_traffic(n_img, fused, dtype):
    """HBM bytes per launch of the roofline kernel from the committed PMC passes (profiles/r06_fused_traffic.json:
    FETCH_SIZE / WRITE_SIZE in separate rocprofv3 --pmc runs of this very command, gfx950 correction applied;
    scratch/gpu_traffic.sh).  The file records the sha256 of the kernel source it was measured on: null when the source
    has changed since (a stale figure must not ride on a new kernel), or when this run's launch is not the measured one."""
    try:
        with open(os.path.join(ROOT, "profiles", "r06_fused_traffic.json")) as f:
            doc = json.load(f)
        m = doc["bench_launch"]
        if doc.get("encoder_fused_hip_sha256") != kernel_source_hash():
            return None
    except (OSError, KeyError, ValueError):
        return None
    if fused and dtype == "bf16" and m["images_per_launch"] == n_img:
        return m["hbm_bytes_per_launch"]
    return None


def cpu_baseline(mod, batch_cpu, noise_cpu, B, budget_s=25.0):
    """The oracle's reference-faithful schedule ((24+12n)B + 16B encoder images, as the reference
    executes them) timed on this host's cores.  Reported baseline, not the target."""
    from oracle import tacorl_oracle as O

    cams = ["rgb_static"]
    spec = O.ACSpec(cams=cams, goal_cams=cams, action_dim=16, n=4, discount=0.95, actor_lr=1e-4, critic_lr=3e-4,
                    deterministic_backup=True, reward_scale=10.0, bc_epochs=5, with_lagrange=True,
                    discrete_gripper=False, target_entropy=-7.0, finetune_action_decoder=False, ac_cams=cams, pr_cams=cams)
    P = {k: v.detach().cpu().clone().contiguous() for k, v in mod.state_dict().items() if v.dtype == torch.float32
         and not any(k.endswith(s) for s in ("one_hot_embedding_eye", "ones", "gripper_bounds", "action_max_bound",
                                               "action_min_bound"))}
    O.require_grad_(P, frozen_prefixes=("perceptual_encoder.", "plan_recognition."))
    opts = O.make_opts(P, spec)
    cores = torch.get_num_threads()
    t0 = time.perf_counter()
    O.tacorl_step(P, opts, spec, batch_cpu, noise_cpu, 5, faithful=True)  # warm-up (allocator, threads)
    first = time.perf_counter() - t0
    reps = max(1, min(5, int(budget_s / max(first, 1e-3)) - 1))
    t0 = time.perf_counter()
    for _ in range(reps):
        O.tacorl_step(P, opts, spec, batch_cpu, noise_cpu, 5, faithful=True)
    dt = (time.perf_counter() - t0) / reps
    return {"value": round(1.0 / dt, 4), "unit": "grad-steps/s (batch 256)", "cores": cores, "kind": "port",
            "sample": f"{reps} full training steps at batch {B} (T=16, 84x84, n=4, Q phase), reference schedule "
                      f"(88*B encoder images/step), after 1 warm-up step; torch-CPU fp32, {cores} threads",
            "s_per_step": round(dt, 3)}


def segment_probe(dtype):
    """`--probe segments` (a child process of the default run; also runnable by hand): the per-GPU step SHAPE of an N-GPU
    run, timed on this one GPU with RCCL really in the loop.  torch.distributed is initialised with backend nccl (= RCCL)
    and world_size 1 and TACORL_FORCE_COLLECTIVES=1 makes every collective of a data-parallel step execute (an all-reduce
    over one rank is the identity): all-reduce #1 (d log alpha) and #2 (the gradient arena) eager between the step's three
    hipGraph segments - the default N-GPU form - and, second, captured as nodes of the step's one graph
    (TACORL_GRAPH_COLLECTIVES=1).  Same process, same module, alternating with the collective-free single graph, so the three
    numbers are a same-box A/B.  Shapes: C2 at B=256 (the weak-scaling share) and C3 (decoder fine-tuning on) at B=32
    (its strong-scaling share of a global batch of 256 on 8 GPUs).  Prints one JSON object."""
    import torch.distributed as dist

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from tacorl_amd import _lib

    _lib.call("tacorl_hip_init", 0)
    warm = torch.ones(1024, device=dev)
    dist.all_reduce(warm)
    torch.cuda.synchronize()
    with open("/proc/self/maps") as f:
        rccl = sorted({ln.split()[-1] for ln in f if "librccl" in ln})

    def timeit(mod, batch, n):
        for _ in range(12):
            mod.training_step(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            mod.training_step(batch)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def forms(mod, batch, n):
        out = {}
        best = {}
        for rep in range(2):  # alternate the forms twice; report the better pass of each (the box drifts by 1-2 %)
            for name, force, ing in (("single_graph_no_collectives", "0", "0"), ("three_segments_eager_rccl", "1", "0"),
                                     ("one_graph_rccl_nodes", "1", "1")):
                if out.get(name) == "failed":
                    continue
                os.environ["TACORL_FORCE_COLLECTIVES"], os.environ["TACORL_GRAPH_COLLECTIVES"] = force, ing
                mod._graphs = {}
                try:
                    ms = timeit(mod, batch, n)
                    best[name] = min(best.get(name, 1e9), ms)
                    out[name] = round(best[name], 4)
                    out.setdefault("graphs_per_step", {})[name] = [len(v[0]) + (v[1] is not None) for v in mod._graphs.values()]
                except Exception as e:  # noqa: BLE001 - a capture that RCCL refuses must not take the other forms down
                    out[name] = "failed"
                    out[name + "_error"] = f"{type(e).__name__}: {str(e)[:300]}"
                    torch.cuda.synchronize()
        os.environ["TACORL_FORCE_COLLECTIVES"], os.environ["TACORL_GRAPH_COLLECTIVES"] = "0", "0"
        mod._graphs = {}
        return out

    res = {"rccl_libraries_mapped": rccl, "backend": dist.get_backend(), "world_size": 1}
    mod = build_module(dev, dtype, 16, 1)
    mod.enable_graph()
    mod.log_every_n_steps = 50
    res["c2_b256"] = forms(mod, synth_batch(256, 16, 84, 84, dev, DATA_SEED), 300)
    mod._graphs = {}
    del mod
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    mod = build_module(dev, dtype, 16, 1, finetune=True)
    mod.enable_graph()
    mod.log_every_n_steps = 50
    res["c3_b32"] = forms(mod, synth_batch(32, 16, 84, 84, dev, DATA_SEED), 300)
    print("PROBE " + json.dumps(res), flush=True)
    dist.destroy_process_group()


def run_segment_probe(dtype, timeout=420):
    """The probe runs in a child process: it initialises RCCL, which the headline measurement never does on one GPU, and
    a hang or crash in there must not cost the JSON line."""
    cmd = [sys.executable, os.path.abspath(__file__), "--probe", "segments", "--dtype", dtype]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        return {"error": f"timeout after {timeout}s"}
    for ln in out.stdout.splitlines():
        if ln.startswith("PROBE "):
            return json.loads(ln[6:])
    return {"error": f"rc {out.returncode}: {out.stderr[-400:]}"}


# ---------------------------------------------------------------------------------------------------------------------
# N-rank supervisor: the first multi-GPU run must not be lost to a hang.
#
# With --gpus N > 1 the processes that the driver (or the user) started never touch a GPU.  They are SUPERVISORS: each
# starts its rank's WORKER as a fresh child process and watches it.  The workers try the step's collective forms in turn,
#   1. "graph-nodes": RCCL all-reduces captured as nodes of the step's ONE hipGraph (fastest; verified with a 1-rank
#      communicator only - the pool hands out 1-GPU boxes),
#   2. "segments":    three collective-free hipGraph segments with eager all-reduces between them (what every 2-rank test runs),
#   3. "eager":       no hipGraph at all,
# and an attempt ends - everywhere - as soon as any rank's worker fails, reports replicas out of sync, or prints nothing
# for TACORL_BENCH_STAGE_TIMEOUT seconds (workers print a heartbeat line per stage).  Then every worker of the attempt is
# killed and the next form starts in NEW processes on a new rendezvous port: a process that has touched the GPU is never
# re-exec'ed or reused.  Two launch modes: `python bench.py --gpus N` (one supervisor, N workers) and
# `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` (N supervisors - torchrun's ranks - that agree
# through a gloo group on host tensors, one worker each).  Rank 0's JSON line is printed once, by its supervisor, with a
# `launcher` block (the attempts and their outcomes) and `config.collectives` naming the form that produced it.
FORMS = (("graph-nodes", {"TACORL_GRAPH_COLLECTIVES": "1"}, ()),
         ("segments", {"TACORL_GRAPH_COLLECTIVES": "0"}, ()),
         ("eager", {"TACORL_GRAPH_COLLECTIVES": "0"}, ("--no-graph",)))
FORM_TEXT = {"graph-nodes": "rccl nodes inside the step's one hipGraph",
             "segments": "eager all-reduces between three hipGraph segments",
             "eager": "eager all-reduces, no hipGraph"}


def _forms():
    names = os.environ.get("TACORL_BENCH_FORMS")
    if names:
        return [f for f in FORMS if f[0] in names.split(",")]
    if os.environ.get("TACORL_DIST_BACKEND", "nccl") != "nccl":
        return list(FORMS[1:])  # gloo cannot be captured into a graph
    return list(FORMS)


def _free_port():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def heartbeat(stage):
    """Worker side: one line per stage on stdout (a pipe to the supervisor, which times the silence between lines)."""
    if os.environ.get("TACORL_BENCH_WORKER"):
        print(f"HB {stage}", flush=True)


def inject_hang(rank):
    """Test hook: TACORL_BENCH_INJECT_HANG="<form>:<rank>" makes that rank's worker of that attempt stall for ever at the
    point where a stuck collective would (tests/test_dist_cpu.py; never set by the driver)."""
    spec = os.environ.get("TACORL_BENCH_INJECT_HANG")
    if spec and os.environ.get("TACORL_BENCH_WORKER"):
        form, r = spec.rsplit(":", 1)
        if form == os.environ.get("TACORL_BENCH_FORM") and int(r) == rank:
            print(f"[rank {rank}] injected hang in form {form}", file=sys.stderr, flush=True)
            while True:
                time.sleep(3600)


class _Worker:
    """One rank's worker process: stdout through a pipe (heartbeats + the JSON line), stderr passed through."""

    def __init__(self, rank, world, port, form, argv):
        import threading

        name, env_extra, extra_args = form
        env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}  # (the agent's store is not ours)
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TACORL_BENCH_WORKER="1", TACORL_BENCH_FORM=name,
                   PYTHONUNBUFFERED="1", **env_extra)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        self.rank, self.lines, self.stage, self.last = rank, [], "start", time.monotonic()
        self.p = subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv, *extra_args], env=env,
                                  stdout=subprocess.PIPE, text=True, bufsize=1)
        self.t = threading.Thread(target=self._read, daemon=True)
        self.t.start()

    def _read(self):
        for ln in self.p.stdout:
            self.last = time.monotonic()
            ln = ln.rstrip("\n")
            if ln.startswith("HB "):
                self.stage = ln[3:]
            else:
                self.lines.append(ln)

    def state(self, start_timeout, stage_timeout):
        """'run' | 'ok' | 'rc <n>' | 'stalled ...'"""
        code = self.p.poll()
        if code is not None:
            self.t.join(timeout=5)
            return "ok" if code == 0 else f"rc {code} at stage {self.stage}"
        idle = time.monotonic() - self.last
        if idle > (start_timeout if self.stage == "start" else stage_timeout):
            return f"stalled {idle:.0f}s at stage {self.stage}"
        return "run"

    def kill(self):
        if self.p.poll() is None:
            self.p.terminate()
            try:
                self.p.wait(timeout=5)
            except subprocess.TimeoutExpired:
                self.p.kill()
                self.p.wait()


def _timeouts():
    # the first `import torch` on a fresh box pages the image in (1-2 min); after that every stage is seconds
    return (float(os.environ.get("TACORL_BENCH_START_TIMEOUT", "420")), float(os.environ.get("TACORL_BENCH_STAGE_TIMEOUT", "150")))


def _publish(lines, attempts, form):
    for ln in lines:
        if ln.startswith("{"):
            try:
                d = json.loads(ln)
                d["launcher"] = {"attempts": attempts, "form": form,
                                 "note": "supervised launch: workers are fresh child processes per attempt; an attempt ends "
                                         "everywhere when any rank fails or is silent past its stage timeout"}
                ln = json.dumps(d)
            except ValueError:
                pass
        print(ln, flush=True)


def launch_ranks(n, argv):
    """`python bench.py --gpus N`: ONE supervisor (this process, which has touched no GPU) and N workers per attempt."""
    t_start, t_stage = _timeouts()
    attempts = []
    for form in _forms():
        ws = [_Worker(r, n, port, form, argv) for port in [_free_port()] for r in range(n)]
        outcome = None
        while outcome is None:
            time.sleep(0.2)
            st = [w.state(t_start, t_stage) for w in ws]
            bad = [f"rank {w.rank}: {s}" for w, s in zip(ws, st) if s not in ("run", "ok")]
            if bad:
                outcome = "; ".join(bad)
            elif all(s == "ok" for s in st):
                outcome = "ok"
        for w in ws:
            w.kill()
        attempts.append({"form": form[0], "outcome": outcome})
        if outcome == "ok":
            _publish(ws[0].lines, attempts, form[0])
            return 0
        print(f"[bench supervisor] form {form[0]} abandoned: {outcome}", file=sys.stderr, flush=True)
    return 1


def supervise_rank(argv):
    """Under torch.distributed.run: this process is rank RANK's supervisor.  The supervisors form a gloo group on host
    tensors (the rendezvous torchrun set up; no GPU call) and agree once a second on the attempt's fate."""
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    t_start, t_stage = _timeouts()
    attempts, rc = [], 1
    for form in _forms():
        port = torch.tensor([_free_port() if rank == 0 else 0])
        dist.broadcast(port, 0)
        w = _Worker(rank, world, int(port.item()), form, argv)
        outcome = None
        while outcome is None:
            time.sleep(1.0)
            s = w.state(t_start, t_stage)
            v = torch.tensor([float(s == "ok"), float(s not in ("run", "ok"))])
            dist.all_reduce(v)  # every supervisor sees the same sums, so every supervisor takes the same decision
            if v[1] > 0:
                outcome = s if s not in ("run", "ok") else "abandoned: another rank failed or stalled"
            elif v[0] == world:
                outcome = "ok"
        w.kill()
        attempts.append({"form": form[0], "outcome": outcome})
        if outcome == "ok":
            if rank == 0:
                _publish(w.lines, attempts, form[0])
            rc = 0
            break
        print(f"[bench supervisor {rank}] form {form[0]} abandoned: {outcome}", file=sys.stderr, flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return rc


def launch_check(world, rank):
    """`--launch-check`: rendezvous + one all-reduce on host tensors, no GPU work (the CPU test of the launcher)."""
    import torch.distributed as dist

    heartbeat("imported")
    dist.init_process_group(os.environ.get("TACORL_DIST_BACKEND", "gloo"))
    heartbeat("group")
    inject_hang(rank)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"launch_check": True, "world": dist.get_world_size(), "sum": t.item()}), flush=True)
    dist.destroy_process_group()


def condition_chip(mod, ms):
    """Keep the chip under load for `ms` right in front of the timed region WITHOUT taking training steps: the step's fused
    encoder-forward launch (engine.encode_fused_only: reads the staged images and the encoders' packed weights, writes the
    embedding / activation buffers the next step overwrites anyway - no parameter, optimiser or noise state changes) back to
    back.  Why: the warm-up steps end behind seconds of host-side work (module construction, the eager pass, the hipGraph
    capture) during which the GPU idles at a low clock; measured on this pool (scratch/first_steps.py) the first steps
    behind 5 warm-up steps then run 0.85, 0.85, 0.84 ... and reach the steady 0.82 ms only after ~25 steps - a 20-step timed
    region reads 0.833 - 0.849 against 0.817 - 0.822 with 100 - 300 ms of conditioning, which is the steady state the 200-step
    default run and the per-step median report.  Returns the ms actually spent (0 when not applicable)."""
    if ms <= 0 or not hasattr(mod, "engine") or not all(mod.engine._fused_ok(c) for c in mod.engine.cams):
        return 0.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(20):
            mod.engine.encode_fused_only()
        torch.cuda.synchronize()
    return round((time.perf_counter() - t0) * 1e3, 1)


def timed_steps(mod, batch, steps, barrier):
    """Contract region: exactly `steps` training steps between barrier + synchronize on both sides."""
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        mod.training_step(batch)
    barrier()
    return time.perf_counter() - t0


def step_time_distribution(mod, batch, ms_guess, min_seconds=1.0, max_steps=4000):
    """Per-step device times from HIP events (one event after every step on the stream the step runs on),
    over at least `min_seconds` of steps: median / p10 / p90 next to the contract region's wall-clock mean."""
    n = int(min(max_steps, max(50, math.ceil(min_seconds * 1e3 / max(ms_guess, 1e-3)))))
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    torch.cuda.synchronize()
    evs[0].record()
    for i in range(n):
        mod.training_step(batch)
        evs[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(n))
    q = lambda f: ts[min(n - 1, int(f * n))]  # noqa: E731
    return {"n": n, "median_ms": round(q(0.5), 4), "p10_ms": round(q(0.1), 4), "p90_ms": round(q(0.9), 4),
            "mean_ms": round(sum(ts) / n, 4), "max_ms": round(ts[-1], 4),
            "how": "hipEvent after every training_step on torch's current stream"}


def time_other_configs(dev, dtype, budget_s=1.0):
    """The other BASELINE configurations at full size on this GPU, ~1 s of timed steps each, so that their numbers sit in
    the driver-timed JSON line instead of in prose: C3 (the default TACORL step: action-decoder fine-tuning on, B=256),
    C4's per-GPU share (dual camera 128x128, window 32, latent 32, B = 512 / 8), C5 (CQL_Offline, discrete gripper, 32
    action samples, B=1024) and C1's module (PlayLMP.training_step at B=32 and B=256).  hipGraph on, metrics read back
    every 50 steps; `enc_frac` = the step's encoder-forward launch against the bf16 MFMA peak (13.918 MFLOP per 84x84
    image; 128x128: 35.35 MFLOP)."""
    from tacorl_amd import synth
    from tacorl_amd.modules.cql.cql_offline_lightning import CQL_Offline
    from tacorl_amd.modules.play_lmp.play_lmp_for_rl import PlayLMP
    from tacorl_amd.modules.tacorl.tacorl import TACORL

    actor = {"policy": {"num_layers": 3, "hidden_dim": 256}}
    critic = {"q_network": {"num_layers": 3, "hidden_dim": 256, "last_layer_activation": "Identity"}}
    yaml = dict(action_decoder_lr=3e-4, actor_lr=1e-4, critic_lr=3e-4, discount=0.95, conservative_weight=1.0,
                reward_scale=10.0, n_action_samples=4, with_lagrange=True, deterministic_backup=True, bc_epochs=5)
    peak = PEAK_BF16_TFLOPS if dtype == "bf16" else PEAK_F32_TFLOPS

    def lmp(cams, T, latent):
        pr = dict(num_heads=8, num_layers=2, encoder_hidden_size=2048, fc_hidden_size=4096, latent_plan_dim=latent,
                  min_std=1e-4, dropout_p=0.0, max_position_embeddings=T)
        ad = dict(n_mixtures=10, num_layers=2, hidden_size=2048, out_features=7, num_classes=10, latent_plan_dim=latent,
                  rnn_model="rnn_decoder", include_goal=False)
        torch.manual_seed(PARAM_SEED)
        return PlayLMP(plan_proposal=actor, plan_recognition=pr, action_decoder=ad, plan_proposal_obs_modalities=cams,
                       plan_proposal_goal_modalities=cams, plan_recognition_modalities=cams, action_decoder_modalities=cams,
                       real_world=True, device=dev, compute_dtype=dtype, image_dtype=dtype)

    def play_batch(B, T, cams):
        g = torch.Generator(device=dev).manual_seed(DATA_SEED + 7)
        u = lambda *s: torch.rand(*s, device=dev, generator=g) * 2 - 1  # noqa: E731
        acts = u(B, T, 7)
        acts[..., -1] = torch.where(acts[..., -1] >= 0, 1.0, -1.0)
        disp = torch.empty(B, device=dev).geometric_(0.3, generator=g).long()
        return {"states": {c: u(B, T, 3, h, w) for c, (h, w) in cams.items()}, "goal": {c: u(B, 3, h, w) for c, (h, w) in cams.items()},
                "actions": acts, "disp": disp}

    def run(mod, batch, args, B, flop_per_img=None):
        mod.enable_graph()
        mod.log_every_n_steps = 50
        for _ in range(5):
            mod.training_step(batch, *args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            mod.training_step(batch, *args)
        torch.cuda.synchronize()
        est = (time.perf_counter() - t0) / 10
        n = int(min(max(budget_s / est, 20), 2000))
        t0 = time.perf_counter()
        for _ in range(n):
            mod.training_step(batch, *args)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        res = {"ms_per_step": round(ms, 4), "samples_per_s": round(B / ms * 1e3, 1), "steps": n, "batch": B}
        eng = getattr(mod, "engine", None)
        if eng is not None and flop_per_img and all(eng._fused_ok(c) for c in eng.cams):
            for _ in range(5):
                n_img = eng.encode_fused_only()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(50):
                eng.encode_fused_only()
            ev1.record()
            torch.cuda.synchronize()
            ems = ev0.elapsed_time(ev1) / 50
            if isinstance(flop_per_img, dict):  # cameras of different geometries (one fused launch each): FLOPs summed per camera
                flop = sum(sum(x[4] for x in eng._all_problems(c)) * flop_per_img[c] for c in eng.cams)
            else:
                flop = n_img * flop_per_img
            res.update(enc_ms=round(ems, 4), enc_images=n_img, enc_frac=round(flop / (ems * 1e-3) / 1e12 / peak, 4))
        fin = getattr(eng, "logs", getattr(mod, "logs", None))
        res["losses_finite"] = bool(torch.isfinite(fin).all().item()) if fin is not None else None
        return res

    def release(mod):
        # drop the module's captured graphs NOW: a process that accumulates dozens of live instantiated hipGraphs
        # segfaults in hipGraphLaunch on ROCm 7.2 (the test suite does the same after every test)
        import gc

        torch.cuda.synchronize()
        mod._graphs = {}
        del mod
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()

    out = {}
    cams84, cams128 = {"rgb_static": (84, 84)}, {"rgb_gripper": (128, 128), "rgb_static": (128, 128)}
    # experiment=tacorl_real_world as the reference configures it: rgb_static un-resized at 150 x 200
    # (config/datamodule/transform_manager/transforms/rl_real_world_train.yaml:2-10), rgb_gripper 84 x 84
    cams_rw = {"rgb_gripper": (84, 84), "rgb_static": (150, 200)}

    def tacorl(cams, T, latent, finetune):
        m = TACORL(play_lmp=lmp(sorted(cams), T, latent), finetune_action_decoder=finetune, critic=critic, real_world=True,
                   device=dev, compute_dtype=dtype, image_dtype=dtype, **yaml)
        m.current_epoch = 5
        return m

    def cql():
        m = CQL_Offline(actor=dict(actor, discrete_gripper=True), critic=critic, real_world=True, obs_modalities=["rgb_static"],
                        goal_modalities=["rgb_static"], action_dim=7, device=dev, compute_dtype=dtype, image_dtype=dtype,
                        discount=0.99, actor_lr=1e-4, critic_lr=3e-4, conservative_weight=1.0, n_action_samples=32,
                        with_lagrange=True, reward_scale=10.0, deterministic_backup=False, bc_epochs=5)
        m.current_epoch = 5
        return m

    to_dev = lambda x: {k: to_dev(v) for k, v in x.items()} if isinstance(x, dict) else (x.to(dev) if torch.is_tensor(x) else x)  # noqa: E731
    plan = [
        ("c3_tacorl_finetune_b256", lambda: tacorl(cams84, 16, 16, True), lambda: play_batch(256, 16, cams84), (), 256, ENC_FLOP_PER_IMG_84),
        # 2 * 17 676 288 MAC per 128 x 128 image
        ("c4_share_dualcam128_b64", lambda: tacorl(cams128, 32, 32, False), lambda: play_batch(64, 32, cams128), (), 64, 35.353e6),
        # 150 x 200: 2 * 35 303 424 MAC per image (encoder_ring.hip); enc_ms = both cameras' fused launches
        ("c4_real_150x200_b64", lambda: tacorl(cams_rw, 32, 32, False), lambda: play_batch(64, 32, cams_rw), (), 64,
         {"rgb_gripper": ENC_FLOP_PER_IMG_84, "rgb_static": 70.607e6}),
        ("c5_cql_n32_b1024", cql, lambda: to_dev(synth.make_transition_batch(7, 1024, cams84)), (0,), 1024, ENC_FLOP_PER_IMG_84),
        ("c1_playlmp_b32", lambda: lmp(["rgb_static"], 16, 16), lambda: play_batch(32, 16, cams84), (0,), 32, None),
        ("c1_playlmp_b256", lambda: lmp(["rgb_static"], 16, 16), lambda: play_batch(256, 16, cams84), (0,), 256, None),
    ]
    for name, make, make_batch, args, B, flop in plan:
        m = None
        try:  # (one configuration's failure must not take the others - or the headline line - with it)
            m = make()
            out[name] = run(m, make_batch(), args, B, flop)
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
        if m is not None:
            release(m)
        del m
    torch.cuda.empty_cache()
    return out


def time_feeder(mod, a, B, T, H, W, dev, barrier, max_over_ranks, n_frames=40000):
    """The step fed by the replay data path instead of a resident batch: per step the host samples window / goal
    indices (PlayIndex, the reference's PlayDataset logic), the frames come out of the uint8 dataset (HBM gather, or
    pinned host memory -> staging ring -> H2D on a copy stream, one batch ahead) and are normalised (optionally
    augmented) on the way into the encoder's buffers."""
    import numpy as np

    from tacorl_amd.data.augment import AugmentSpec, draw_play_batch_augmentation
    from tacorl_amd.data.replay import HbmReplay, PinnedReplay, PlayIndex

    g = torch.Generator().manual_seed(5)
    frames = torch.randint(0, 256, (n_frames, H, W, 3), dtype=torch.uint8, generator=g)
    acts = np.random.RandomState(6).uniform(-1, 1, size=(n_frames, 7)).astype(np.float32)
    ix = PlayIndex([[i, i + 1999] for i in range(0, n_frames, 2000)], T, T, goal_sampling_prob=0.3)
    rng = np.random.default_rng(7)
    spec = {"rgb_static": AugmentSpec(pad=4)} if a.augment else None
    pinned = a.feeder.startswith("pinned")
    rep = (PinnedReplay({"rgb_static": frames}, acts, ix, dev, gather="host" if a.feeder == "pinned-host" else "device")
           if pinned else HbmReplay({"rgb_static": frames}, acts, ix, dev))

    def draw():
        return rng.integers(len(ix), size=B), ix.draw(B, rng), (draw_play_batch_augmentation(spec, B, T, dev) if spec else None)

    def run(steps, evs=None):
        def mark():
            if evs is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                evs.append(e)

        mark()
        if pinned:
            rep.prefetch(*draw())
            for _ in range(steps):
                b = rep.next()
                rep.prefetch(*draw())  # the gather of the next batch (over PCIe) overlaps this step
                mod.training_step(b)
                mark()
        else:
            from tacorl_amd.data.replay import prefetching

            # frames by index straight into the encoder's buffers; the host side of a batch is prepared two ahead
            for b in prefetching(lambda: rep.batch(*draw(), fused=True), steps):
                mod.training_step(b)
                mark()

    run(max(a.warmup, 50))  # (the first ~40 fed steps hold one-time host stalls of ~13 ms each: capture of the new batch key,
    barrier()               # pinned / device staging buffers, the prefetch thread's first allocations)
    evs = []
    t0 = time.perf_counter()
    run(a.steps, evs)
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    ts = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(len(evs) - 1))
    q = lambda f: round(ts[min(len(ts) - 1, int(f * len(ts)))], 4)  # noqa: E731
    return {"kind": a.feeder, "augment": bool(a.augment), "ms_per_step": round(dt / a.steps * 1e3, 4),
            "p50_ms": q(0.5), "p90_ms": q(0.9), "p99_ms": q(0.99), "max_ms": round(ts[-1], 4),
            "steps_per_s": round(a.steps / dt, 2), "dataset_frames": n_frames,
            "bytes_per_step_uint8": int(B * (T + 1) * H * W * 3),
            "note": "hbm: nothing but indices/actions crosses PCIe; pinned: PCIe-inclusive, the GPU gathers the frames out of pinned "
                    "host memory one batch ahead; pinned-host: host gather into a staging ring + H2D copy"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-distribution", action="store_true", help="skip the per-step HIP-event statistics")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the ~10 s `configs` block (C3 / C4 share / C5 / PlayLMP / hbm feeder after the headline region)")
    ap.add_argument("--launch-check", action="store_true", help="rendezvous + one host all-reduce only (launcher test)")
    ap.add_argument("--ad-every", type=int, default=1,
                    help="evaluate the (logging-only, frozen) action-decoder loss every k-th step; 1 = every step "
                         "as the reference does")
    ap.add_argument("--feeder", default="none", choices=["none", "hbm", "pinned", "pinned-host"],
                    help="also time the step fed by the replay data path (SURVEY 8f N2/N3): hbm = uint8 dataset resident in HBM, "
                         "windows gathered on the GPU; pinned = dataset in pinned host memory, the GPU gathers the windows out of it over "
                         "PCIe on a copy stream one batch ahead (the PCIe-inclusive number); pinned-host = host gather into a staging "
                         "ring + H2D copy.  Reported in a `feeder` block; `value` is unchanged")
    ap.add_argument("--augment", action="store_true", help="with --feeder: RandomShiftsAug + ColorJitter on the way in")
    ap.add_argument("--frames", default="f32", choices=["f32", "u8"],
                    help="f32: the reference's batch schema (transformed fp32 CHW frames; the contract of `value`); "
                         "u8: the dataset's uint8 HWC frames, normalised on the GPU (SURVEY 8f N2; reported in DESIGN.md)")
    ap.add_argument("--probe", default=None, choices=["segments"], help="internal: run one of the child-process probes")
    ap.add_argument("--condition-ms", type=float, default=150.0,
                    help="chip conditioning between the warm-up steps and the timed region: the step's encoder-forward launch back "
                         "to back for this many ms (no training step, no parameter changes; 0 = none).  Behind the host-side "
                         "capture the chip clocks up over ~25 steps: without it a 20-step region reads 2-3 %% above the steady state")
    a = ap.parse_args()

    if a.probe == "segments":
        return segment_probe(a.dtype)
    if a.gpus > 1 and not os.environ.get("TACORL_BENCH_WORKER"):
        # no GPU call has happened (or will happen) in this process: it supervises fresh worker processes
        if "WORLD_SIZE" not in os.environ:
            sys.exit(launch_ranks(a.gpus, sys.argv[1:]))
        if int(os.environ["WORLD_SIZE"]) != a.gpus:
            sys.exit(f"--gpus {a.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: start with --nproc-per-node {a.gpus}")
        sys.exit(supervise_rank(sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        sys.exit(f"--gpus {a.gpus} but WORLD_SIZE={world}: start with --nproc-per-node {a.gpus}, or without "
                 f"torch.distributed.run (bench.py then starts its own rank processes)")
    if a.launch_check:
        return launch_check(world, rank)
    # Test hooks (never set by the driver): run the N-rank code path on a 1-GPU box - every rank on
    # cuda:0 and gloo instead of RCCL (which refuses two ranks on one device).
    if os.environ.get("TACORL_BENCH_SINGLE_DEVICE"):
        local = 0
    backend = os.environ.get("TACORL_DIST_BACKEND", "nccl")
    heartbeat("imported")
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":  # = RCCL on ROCm
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        heartbeat("group")

    from tacorl_amd import _lib

    _lib.call("tacorl_hip_init", local)
    B, T, H, W = a.batch, 16, 84, 84
    # identical parameters on every rank: same PARAM_SEED, and the constructor broadcasts rank 0's blocks
    # (what DDP's wrap does for the reference module)
    mod = build_module(dev, a.dtype, T, world, a.ad_every)
    # every rank draws its own noise for its own shard of the global batch
    torch.manual_seed(DATA_SEED + rank)
    torch.cuda.manual_seed(DATA_SEED + rank)
    batch = synth_batch(B, T, H, W, dev, DATA_SEED + rank)
    if a.frames == "u8":  # the same frames quantised back to the dataset's format
        q = lambda x, perm: ((x.permute(*perm) * 0.5 + 0.5) * 255).round().clamp(0, 255).to(torch.uint8).contiguous()  # noqa: E731
        batch = dict(batch, states={k: q(v, (0, 1, 3, 4, 2)) for k, v in batch["states"].items()},
                     goal={k: q(v, (0, 2, 3, 1)) for k, v in batch["goal"].items()})
    use_graph = (not a.no_graph) and hasattr(mod, "enable_graph")
    if use_graph:
        mod.enable_graph()
    mod.log_every_n_steps = 50  # PL Trainer default: metrics are read back every 50 steps

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world > 1:
            t = torch.tensor([x], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return t.item()
        return x

    from tacorl_amd import dist as D

    heartbeat("module built")
    ranks_seen = None
    if world > 1:
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)  # (also the communicator's first collective, outside any capture)
        ranks_seen = int(ones.item())
        heartbeat("first all-reduce")
    for i in range(max(a.warmup, 2)):
        mod.training_step(batch)
        if i == 1:  # eager pass + capture done, first replay issued
            torch.cuda.synchronize()
            heartbeat("captured")
            inject_hang(rank)
    torch.cuda.synchronize()
    heartbeat("warm")
    coll_form = None
    if world > 1:  # read AFTER the captures: a refused in-graph capture falls back to segments inside the library
        coll_form = FORM_TEXT["eager"] if not use_graph else FORM_TEXT["graph-nodes" if D.graph_collectives() else "segments"]
        if os.environ.get("TACORL_BENCH_FORM") == "graph-nodes" and not D.graph_collectives():
            coll_form += " (in-graph capture was refused)"
    # (VERDICT r5 / advisor: the same K steps right behind the warm-up, WITHOUT conditioning, reported beside the headline)
    uncond_ms = None
    if a.condition_ms > 0:
        uncond_ms = round(max_over_ranks(timed_steps(mod, batch, a.steps, barrier)) / a.steps * 1e3, 4)
        heartbeat("timed unconditioned")
    cond_ms = condition_chip(mod, a.condition_ms)
    heartbeat("conditioned")
    my_dt = timed_steps(mod, batch, a.steps, barrier)
    dt = max_over_ranks(my_dt)
    ms_step = dt / a.steps * 1e3
    rank_ms = None
    if world > 1:  # arrival skew: every rank's own wall time over the same region
        lo = torch.tensor([my_dt], device=dev, dtype=torch.float64)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        rank_ms = {"min": round(lo.item() / a.steps * 1e3, 4), "max": round(ms_step, 4)}
    heartbeat("timed")
    logs = mod.engine.metrics()
    finite = all(v == v and abs(v) < 1e30 for v in logs.values())
    dist_stats = None if a.no_distribution else step_time_distribution(mod, batch, ms_step)
    # the roofline probe of the dominant kernel, taken here - straight behind the timed steps, the chip in the clock state
    # the step runs in (behind the other blocks below, after a module teardown, the same launch read up to 8 % slower)
    enc_probe = time_encoder_fwd(mod, B, H, W) if rank == 0 else None
    # (training steps: on N ranks they hold the step's all-reduces, so every rank runs them; rank 0's reading is reported)
    enc_in_step = time_encoder_in_step(mod, batch)
    heartbeat("probes")

    def replicas_in_sync():
        e = mod.engine
        cs = torch.stack([x.param.double().sum() for x in (e.actor, e.q1, e.q2)]).to(dev)
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        return bool(torch.equal(lo, hi))

    in_sync, strong = None, None
    if world > 1:  # replicas must hold identical parameters after the timed steps (same all-reduced grads)
        in_sync = replicas_in_sync()
        if not in_sync and os.environ.get("TACORL_BENCH_WORKER"):
            # a form whose all-reduces did not reduce is a failed attempt: the supervisor moves on to the next one
            sys.exit(f"[rank {rank}] replicas out of sync after the timed steps with {coll_form}")
        # strong scaling: the SAME global batch of 256, 256/N samples per GPU -> true optimiser steps per second
        if B % world == 0:
            sb = synth_batch(B // world, T, H, W, dev, DATA_SEED + 100 + rank)
            for _ in range(a.warmup):
                mod.training_step(sb)
            sdt = max_over_ranks(timed_steps(mod, sb, a.steps, barrier))
            strong = {"global_batch": B, "per_gpu_batch": B // world, "ms_per_step": round(sdt / a.steps * 1e3, 4),
                      "value": round(a.steps / sdt, 3), "unit": "optimiser steps/s at global batch 256",
                      "replicas_in_sync": replicas_in_sync()}
            for _ in range(3):  # back to the bench shape for the roofline probe below
                mod.training_step(batch)
        heartbeat("strong")

    feeder = None
    if a.feeder != "none":
        feeder = time_feeder(mod, a, B, T, H, W, dev, barrier, max_over_ranks)
        for _ in range(3):
            mod.training_step(batch)
    configs = None
    if world == 1 and not a.no_configs and a.frames == "f32" and not a.launch_check:
        # (after the contract's timed region; `value` is untouched)
        configs = {}
        if a.feeder == "none":
            fa = argparse.Namespace(**{**vars(a), "feeder": "hbm", "steps": max(400, a.steps), "warmup": 10})  # (never the driver's --steps 20: one outlier moved that mean by 9 %)
            f = time_feeder(mod, fa, B, T, H, W, dev, barrier, max_over_ranks)
            configs["c2_fed_from_hbm_replay"] = {k: f[k] for k in ("ms_per_step", "p50_ms", "p90_ms", "p99_ms", "max_ms", "steps_per_s",
                                                                   "dataset_frames", "bytes_per_step_uint8")}
            for _ in range(3):
                mod.training_step(batch)
        try:
            configs.update(time_other_configs(dev, a.dtype))
        except Exception as e:  # the headline line must survive a failure here
            configs["error"] = f"{type(e).__name__}: {e}"
        torch.cuda.synchronize()
        # the per-GPU step shape of an N-GPU run with RCCL executing (1-rank communicator), in a child process
        pr = run_segment_probe(a.dtype)
        c2, c3 = pr.get("c2_b256", {}), pr.get("c3_b32", {})
        configs["c2_three_segment"] = dict(c2, rccl=pr.get("rccl_libraries_mapped"), error=pr.get("error"),
                                           note="ms/step at B=256: the collective-free single graph (= the headline form), 3 graph "
                                                "segments with 2 eager RCCL all-reduces between them (the only form gloo can run) and the "
                                                "one-graph form with the all-reduces as graph nodes (the N-GPU default with backend nccl); "
                                                "world_size 1, same process")
        configs["c3_strong_share_b32"] = dict(c3, note="C3 (decoder fine-tuning) at B=32 = its per-GPU share of a global batch of "
                                                       "256 on 8 GPUs, same three forms")
        for _ in range(3):
            mod.training_step(batch)

    out = None
    if rank == 0:
        b2b_ms, n_img, fused = enc_probe
        enc_ms = enc_in_step[0] if (fused and enc_in_step is not None) else b2b_ms
        tflops = n_img * ENC_FLOP_PER_IMG_84 / (enc_ms * 1e-3) / 1e12
        peak = PEAK_BF16_TFLOPS if a.dtype == "bf16" else PEAK_F32_TFLOPS
        out = {
            "metric": "offline grad-steps/sec (batch=256, 84x84 RGB)",
            "value": round(world * (B / 256.0) / (ms_step * 1e-3), 3),
            "unit": "grad-steps/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": "tacorl actor-critic training_step, frozen LMP (BASELINE configs[1])",
                       "per_gpu_batch": B, "global_batch": B * world, "window": T, "image": f"{H}x{W}x3",
                       "frames": "fp32 CHW (reference batch schema)" if a.frames == "f32" else "uint8 HWC, normalised on the GPU",
                       "latent_plan": 16, "n_action_samples": 4, "phase": "Q (epoch>=bc_epochs)",
                       "action_decoder_loss": ("every step (reference behaviour)" if a.ad_every <= 1 else
                                               f"every {a.ad_every} steps (logging cadence)"), "parallelism": f"dp{world}",
                       "collective_backend": None if world == 1 else ("rccl" if backend == "nccl" else backend),
                       "collectives": coll_form, "rccl_ranks_seen": ranks_seen, "rank_ms_per_step": rank_ms,
                       "hip_graph": bool(use_graph),
                       "chip_conditioning": {"ms": cond_ms, "unconditioned_ms": uncond_ms,
                                             "what": "the step's encoder-forward launch back to back between the warm-up "
                                             "steps and the timed region (no training step, no state change): the timed steps "
                                             "then run at the steady-state clock instead of ramping up over their first ~25; "
                                             "unconditioned_ms = the same number of steps timed right behind the warm-up, before "
                                             "any conditioning"},
                       "samples_per_s": round(world * B / (ms_step * 1e-3), 1), "losses_finite": finite,
                       "replicas_in_sync": in_sync},
            "step_time": dist_stats,
            "strong": strong,
            "feeder": feeder,
            "configs": configs,
            "roofline": {"bound": "mfma", "achieved": round(tflops, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(tflops / peak, 4), "traffic": measured_traffic(n_img, fused, a.dtype),
                         "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/)",
                         "kernel": ("encoder_fused_kernel: the step's single LMPVisionEncoder forward launch "
                                    "(all 27*B encoder images: frozen LMP B*T frames, actor/q1/q2 over "
                                    "[obs;goal], actor(next), both targets)") if fused else
                                   "LMPVisionEncoder forward, per-layer kernels (tacorl_encoder_fwd)",
                         "images_per_launch": n_img, "avg_ms": round(enc_ms, 4),
                         "how": ("HIP events around the launch inside 40 training steps (eager replays of the step), minus the "
                                 "bracket's launch gaps, measured on a 120 us one-thread spin kernel that records its own "
                                 "duration, bracketed right behind it"
                                 if (fused and enc_in_step is not None) else "HIP events over 100 back-to-back launches"),
                         "event_bracket_ms": round(enc_in_step[1], 4) if enc_in_step else None,
                         "event_overhead_ms": round(enc_in_step[2], 4) if enc_in_step else None,
                         "back_to_back_ms": round(b2b_ms, 4)},
        }
        if world == 1 and not a.no_cpu_baseline and a.frames == "f32":
            bc = {"states": {"rgb_static": batch["states"]["rgb_static"].cpu()},
                  "goal": {"rgb_static": batch["goal"]["rgb_static"].cpu()}, "actions": batch["actions"].cpu(),
                  "disp": batch["disp"].cpu()}
            nz = {k: v.cpu().clone() for k, v in mod.engine.noise.items()}
            nz["eps_pr"] = mod.eps_pr.cpu().clone()
            out["cpu_baseline"] = cpu_baseline(mod, bc, nz, B)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
