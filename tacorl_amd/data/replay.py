"""Play-data replay for the offline step: the window / hindsight-goal sampling of the reference's
PlayDataset (datamodule/dataset/play_dataset.py:115-169,258-310,452-473) as index arithmetic over frame ids, and two
frame stores that turn the ids into the module's uint8 batch:

* HbmReplay    - MI355X-first: the whole uint8 dataset lives in HBM (84x84x3 = 21 KB per frame; 2.4 M CALVIN frames
                 = 50 GB of 288 GB), a step's frames are gathered on the GPU (`tacorl_gather_frames_u8`), nothing
                 crosses PCIe per step but a few KB of indices and actions;
* PinnedReplay - the dataset in pinned host memory, gathered by the host into a double-buffered pinned staging
                 ring and copied H2D on a copy stream one batch ahead (the PCIe-inclusive path).

Every random draw is explicit (`PlayIndex.draw`), so the sampling is a deterministic function that
tests/golden/play_sampler.npz (recorded from the reference class) pins.
"""
import numpy as np
import torch

from .. import ops

GEOMETRIC, SIMILAR = 0, 1


class PlayIndex:
    def __init__(self, ep_start_end_ids, min_window_size=16, max_window_size=16, goal_sampling_prob=0.3,
                 goal_strategy_prob=None, goal_augmentation=False, nn_steps_from_step=None, train=True):
        if min_window_size > max_window_size:
            raise ValueError(f"min_window_size {min_window_size} > max_window_size {max_window_size}")  # :133-137
        self.ep = np.asarray(ep_start_end_ids, dtype=np.int64).reshape(-1, 2)
        self.min_ws, self.max_ws, self.train = int(min_window_size), int(max_window_size), train
        self.p_geom_goal = float(goal_sampling_prob)
        gs = goal_strategy_prob or {"geometric": 0.5, "similar_robot_obs": 0.5}
        if not np.isclose(sum(gs.values()), 1.0):
            raise ValueError("Goal strategy probability must sum to 1")  # :93-96
        self.p_strategy = np.array([gs.get("geometric", 0.0), gs.get("similar_robot_obs", 0.0)])
        self.goal_augmentation = bool(goal_augmentation)
        self.nn = {int(k): np.asarray(v, dtype=np.int64) for k, v in (nn_steps_from_step or {}).items()}
        # the ragged neighbour lists as one CSR table indexed by frame id (a python loop over half the batch was the
        # sampler's hot spot: 180 us per 256-sample batch, a fifth of the device step)
        top = (max(self.nn) + 1) if self.nn else 0
        cnt = np.zeros(top, dtype=np.int64)
        for k, v in self.nn.items():
            cnt[k] = len(v)
        self._nn_ptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
        self._nn_val = (np.concatenate([self.nn[k] for k in sorted(self.nn)]) if self.nn else np.zeros(0, np.int64))
        # load_file_indices (:452-473): every start frame that leaves room for a max-size window
        look = []
        for s, e in self.ep:
            assert e > self.max_ws
            look.append(np.arange(s, e + 1 - self.max_ws))
        self.episode_lookup = np.concatenate(look) if look else np.zeros(0, np.int64)

    def __len__(self):
        return len(self.episode_lookup)

    def draw(self, n, rng):
        """The random quantities one batch of n items consumes, as the reference draws them per item:
        window size ~ randint(min, max+1) (:121-124), strategy ~ choice(p) (:171-174), disp ~ geometric(p) (:270),
        noise_step ~ randint(3)-1 (:273), and two uniforms that pick the similar-robot-obs goal / the random state."""
        return {"window_size": rng.integers(self.min_ws, self.max_ws + 1, size=n),
                "strategy": (rng.random(n) >= self.p_strategy[0]).astype(np.int64),
                "disp": rng.geometric(self.p_geom_goal, size=n), "noise_step": rng.integers(0, 3, size=n) - 1,
                "u_choice": rng.random(n), "u_random_state": rng.random(n)}

    def episode_end(self, step):
        """find_episode_end (:238-242): end (inclusive) of the first episode with start <= step <= end, -1 if none."""
        step = np.asarray(step)
        k = np.searchsorted(self.ep[:, 0], step, side="right") - 1
        ok = (k >= 0) & (step <= self.ep[np.maximum(k, 0), 1])
        return np.where(ok, self.ep[np.maximum(k, 0), 1], -1)

    def sample(self, idx, d):
        """idx (n,) dataset indices; d = draw(...).  Returns frame ids (n, max_ws) padded by repetition of the last
        frame (:300-306), the pad mask, window sizes, the goal frame id and `disp` (geometric k, or -1)."""
        idx = np.asarray(idx, dtype=np.int64)
        n = len(idx)
        ws = np.full(n, self.max_ws) if self.min_ws == self.max_ws else np.asarray(d["window_size"])
        start = self.episode_lookup[idx]
        t = np.arange(self.max_ws)[None, :]
        frames = start[:, None] + np.minimum(t, ws[:, None] - 1)
        padded = t >= ws[:, None]
        # goal (:159-169): geometric future state of the same episode, or a frame with a similar robot state
        strat = np.asarray(d["strategy"])
        disp = np.asarray(d["disp"]).astype(np.int64)
        end = self.episode_end(start)
        goal_step = start + (ws - 1) * disp
        if self.goal_augmentation:
            goal_step = goal_step + np.asarray(d["noise_step"])
        random_state = self.episode_lookup[np.minimum((np.asarray(d["u_random_state"]) * len(self)).astype(np.int64), len(self) - 1)]
        goal = np.where(end >= 0, np.minimum(end, goal_step), random_state)
        out_disp = np.where(strat == GEOMETRIC, disp, -1)
        sim = strat == SIMILAR
        if sim.any():
            key = start + ws - 1
            inside = (key >= 0) & (key < len(self._nn_ptr) - 1)
            kc = np.where(inside, key, 0)
            lo = self._nn_ptr[kc] if len(self._nn_ptr) > 1 else np.zeros(n, np.int64)
            cnt = np.where(inside, (self._nn_ptr[kc + 1] - lo) if len(self._nn_ptr) > 1 else 0, 0)
            pick = np.minimum((np.asarray(d["u_choice"]) * cnt).astype(np.int64), np.maximum(cnt - 1, 0))
            nbr = self._nn_val[np.minimum(lo + pick, max(len(self._nn_val) - 1, 0))] if len(self._nn_val) else random_state
            goal = np.where(sim, np.where(cnt > 0, nbr, random_state), goal)
        # frame ids feed an unchecked device gather: keep them inside the dataset (a goal-augmentation noise step of -1 on a
        # one-frame window at frame 0 is the one way the reference arithmetic leaves it)
        goal = np.clip(goal, 0, int(self.ep[:, 1].max()))
        return {"frames": frames, "padded": padded, "window_size": ws, "goal": goal, "disp": out_disp, "idx": idx}


def check_ids(ids, n_frames):
    """Host-side range check of the frame ids a gather kernel will dereference (it has none of its own)."""
    ids = np.asarray(ids)
    if ids.size and (int(ids.min()) < 0 or int(ids.max()) >= n_frames):
        raise IndexError(f"replay: frame id outside [0, {n_frames}): min {int(ids.min())}, max {int(ids.max())}")


def pad_actions(actions, frames, padded):
    """pad_sequence for the relative actions (:289-297): padded steps are zero except the gripper action, which
    repeats the last real step's.  actions (N,7) host array -> (n, T, 7) float32."""
    a = np.asarray(actions, dtype=np.float32)[frames]
    a[..., :-1][padded] = 0.0
    return a


class _PinnedRing:
    """Small host tensors (ids, actions, flags) go to the device through pinned staging slots and a private COPY stream.
    * pageable source: the copy blocks the host until the stream has reached it, i.e. for the whole step still running;
    * the compute stream itself: with ~10 steps of launches queued ahead of the GPU, hipMemcpyAsync (and the next
      hipGraphLaunch behind it) block inside the runtime until the stream has drained - 8-14 ms stalls of both threads every
      20-100 steps (stacks dumped in the stall: the producer in `copy_`, the trainer in `graph.replay`), +7 % on the
      replay-fed step.  On its own stream the copy only waits, ON THE GPU, for the compute stream's position at the time
      the slot is taken (every step that read this slot's device buffers has been enqueued by then), and the consumer's
      stream waits for the slot's `ready` event (`wait_ready`).
    Each slot is reused only after the copies issued from it have completed (the same event, on the host)."""

    def __init__(self, device, slots=6, slot_bytes=1 << 20):
        self.dev, self.slots, self.k = device, [dict() for _ in range(slots)], 0
        self.events = [None] * slots
        self.stream = torch.cuda.Stream(device=device) if torch.device(device).type == "cuda" else None
        # every slot's pinned + device staging buffer exists before the first batch (1 MiB holds the tables of ~4 000
        # windows of 16 frames): a feeder thread that calls hipHostMalloc / hipMalloc while the trainer thread is inside a
        # hipGraph capture can invalidate that capture (they grow - under ops.capture_lock - only for larger batches)
        if self.stream is not None:
            for sl in self.slots:
                sl["_packed"] = (torch.empty(slot_bytes, dtype=torch.uint8).pin_memory(),
                                 torch.empty(slot_bytes, dtype=torch.uint8, device=self.dev))
        # the stream whose steps read the slots' device buffers: the one current where the ring is BUILT (the trainer's
        # thread) - `begin` may run on a feeder thread, whose own current stream is the default one
        self.consumer = torch.cuda.current_stream(device) if self.stream is not None else None

    def begin(self):
        self.k = (self.k + 1) % len(self.slots)
        if self.events[self.k] is not None:
            self.events[self.k].synchronize()
        if self.stream is not None:
            self.stream.wait_stream(self.consumer)
        return self.k

    def put_all(self, slot, arrays):
        """{key: host array} -> ONE pinned buffer of this slot -> ONE device buffer, moved by a copy KERNEL that reads the
        mapped pinned pages (tacorl_copy_cols over the buffer as 4-byte words), not by hipMemcpyAsync: an async copy on a
        stream that still waits for another stream blocked the producer inside the runtime until that wait had resolved
        (7-8 ms per occurrence); a kernel launch is only queued.  Returns {key: device view}."""
        arrs = {k: np.ascontiguousarray(v) for k, v in arrays.items()}
        offs, off = {}, 0
        for k, a in arrs.items():
            offs[k] = off
            off = (off + a.nbytes + 15) // 16 * 16
        hd = self.slots[slot].get("_packed")
        if hd is None or hd[0].numel() < off:
            from .. import ops

            # growth (a batch larger than the pre-allocated staging): hipHostMalloc / hipMalloc must not run beside a
            # capture, whoever called us - prefetching() holds the lock already (an RLock), a user's own feeder thread
            # calling HbmReplay.batch() directly does not - and captured graphs that could hold the old
            # buffer's address are invalidated
            with ops.capture_lock:
                hd = self.slots[slot]["_packed"] = (torch.empty(off, dtype=torch.uint8).pin_memory(),
                                                   torch.empty(off, dtype=torch.uint8, device=self.dev))
                ops.note_alloc()
        hnp = hd[0].numpy()
        for k, a in arrs.items():
            hnp[offs[k]:offs[k] + a.nbytes] = a.reshape(-1).view(np.uint8)
        if self.stream is not None:
            from .. import ops
            from .._lib import call

            with torch.cuda.stream(self.stream):
                call("tacorl_copy_cols", hd[0].data_ptr(), off // 4, hd[1].data_ptr(), off // 4, 1, off // 4, 0, 0, ops.stream())
        else:
            hd[1].copy_(hd[0])
        return {k: hd[1][offs[k]:offs[k] + a.nbytes].view(torch.from_numpy(a).dtype).view(a.shape) for k, a in arrs.items()}

    def end(self, slot):
        ev = torch.cuda.Event()
        ev.record(self.stream)
        self.events[slot] = ev
        return ev


def wait_ready(batch):
    """Called by the consumer (the modules' frame staging) on the stream that will read the batch: orders that stream
    behind the copies of the batch's small tables (`batch["ready"]`, set by HbmReplay.batch)."""
    ev = batch.get("ready") if isinstance(batch, dict) else None
    if ev is not None:
        torch.cuda.current_stream().wait_event(ev)


class HbmReplay:
    """uint8 HWC frames of every camera resident in HBM: frames[cam] = (N,H,W,3) uint8 device tensor."""

    def __init__(self, frames, actions, index, device=None, consumer_stream=None):
        """consumer_stream: the stream the training steps run on (default: the stream current here, at construction)."""
        self.dev = torch.device(device) if device is not None else next(iter(frames.values())).device
        self.frames = {c: v.to(self.dev).contiguous() for c, v in frames.items()}
        self.actions = np.asarray(actions, dtype=np.float32)
        self.index = index
        self._buf = {}
        self._ring = _PinnedRing(self.dev)
        if consumer_stream is not None:
            self._ring.consumer = consumer_stream

    def _out(self, key, shape):
        t = self._buf.get(key)
        if t is None or t.shape != shape:
            ops.note_alloc()
            t = self._buf[key] = torch.empty(shape, dtype=torch.uint8, device=self.dev)
        return t

    def batch(self, idx, draws, aug=None, fused=False):
        """fused=False: the module's uint8 batch (window frames / goal frames gathered into (n, T, H, W, 3) / (n, H, W, 3)).
        fused=True: no frame is moved here - the batch carries the dataset tensors and the id table
        (`batch["replay"]`), and the module's image pack reads the frames by index straight out of the dataset on their
        way into the encoder's buffers (tacorl_pack_images_u8_gather_batch): one pass instead of gather + pack."""
        s = self.index.sample(idx, draws)
        n, T = s["frames"].shape
        all_ids = np.concatenate([s["frames"].reshape(-1), s["goal"]])
        check_ids(all_ids, min(int(v.shape[0]) for v in self.frames.values()))
        slot = self._ring.begin()
        d = self._ring.put_all(slot, {"ids": all_ids.astype(np.int64, copy=False), "actions": pad_actions(self.actions, s["frames"], s["padded"]),
                                      "disp": s["disp"]})
        ids, acts_d, disp_d = d["ids"], d["actions"], d["disp"]
        ready = self._ring.end(slot)
        if fused:
            b = {"replay": {"frames": self.frames, "ids": ids, "B": n, "T": T}, "actions": acts_d, "disp": disp_d, "ready": ready,
                 "idx": torch.from_numpy(s["idx"]), "window_size": torch.from_numpy(s["window_size"])}
            if aug is not None:
                b["aug"] = aug
            return b
        if ready is not None and self._ring.stream is not None:
            torch.cuda.current_stream(self.dev).wait_event(ready)  # the gathers below read the id table on this stream
        states, goal = {}, {}
        for c, fr in self.frames.items():
            H, W = fr.shape[1:3]
            states[c] = ops.gather_frames_u8(fr, ids[: n * T], self._out(("s", c), (n, T, H, W, 3)))
            goal[c] = ops.gather_frames_u8(fr, ids[n * T:], self._out(("g", c), (n, H, W, 3)))
        b = {"states": states, "goal": goal, "actions": acts_d, "disp": disp_d, "ready": ready, "idx": torch.from_numpy(s["idx"]),
             "window_size": torch.from_numpy(s["window_size"])}
        if aug is not None:
            b["aug"] = aug
        return b


def prefetching(make_batch, n, depth=2):
    """Generator over `n` batches built by `make_batch()` on a background thread, `depth` ahead: the host work of a batch
    (index sampling, action padding, the pinned staging copies: ~0.3 ms for 256 windows) runs beside the previous
    step's launch sequence instead of in front of it - a 0.9 ms device step leaves the main thread no slack for it.
    The producer enqueues its copies on the device's current stream before it hands the batch over, so the step
    that consumes the batch is stream-ordered behind them."""
    import queue
    import threading

    from .. import ops

    q = queue.Queue(maxsize=depth)
    dev = torch.cuda.current_device() if torch.cuda.is_available() else None
    stop = threading.Event()

    def put(x):
        while not stop.is_set():
            try:
                q.put(x, timeout=0.05)
                return True
            except queue.Full:
                pass
        return False

    def produce():
        try:
            if dev is not None:
                torch.cuda.set_device(dev)
            for _ in range(n):
                # the batch's device work (staging-slot copies, events, stream waits) never runs beside a hipGraph capture
                # on the trainer's thread: captures hold the same lock (modules/common.py _run_segments)
                with ops.capture_lock:
                    b = make_batch()
                if not put(b):
                    return
        except BaseException as e:  # surface the failure in the consumer
            put(e)

    th = threading.Thread(target=produce, daemon=True)
    th.start()
    try:
        for _ in range(n):
            b = q.get()
            if isinstance(b, BaseException):
                raise b
            yield b
    finally:
        # the consumer is done (or raised, or dropped the generator early): release a producer blocked on a full queue
        stop.set()
        th.join(timeout=10.0)


class PinnedReplay:
    """The dataset in pinned host memory (datasets larger than HBM).  gather="device" (default): the GPU gathers the
    batch's frames straight out of the pinned pages over PCIe - the same gather kernel as HbmReplay, reading mapped host
    memory - on a copy stream, one batch ahead of the step; the host only draws indices.  gather="host": the host gathers
    into one of two pinned staging buffers and the copy stream moves them (single-threaded index_select: 16 ms per 92 MB
    batch - kept for comparison)."""

    def __init__(self, frames, actions, index, device, gather="device"):
        assert gather in ("device", "host")
        self.dev = torch.device(device)
        self.frames = {c: v.contiguous().pin_memory() for c, v in frames.items()}
        self.actions = np.asarray(actions, dtype=np.float32)
        self.index = index
        self.gather = gather
        self.stream = torch.cuda.Stream(device=self.dev)
        self._slot, self._host, self._dev_buf = 0, [{}, {}], [{}, {}]
        self._copied = [None, None]  # per slot: the copy stream's event behind the last H2D copy out of its staging buffers
        self._pending = None

    def _dev_out(self, slot, key, shape):
        d = self._dev_buf[slot].get(key)
        if d is None or d.shape != shape:
            ops.note_alloc()
            d = self._dev_buf[slot][key] = torch.empty(shape, dtype=torch.uint8, device=self.dev)
        return d

    def _stage(self, slot, key, src, ids):
        shape = (len(ids),) + tuple(src.shape[1:])
        h = self._host[slot].get(key)
        if h is None or h.shape != shape:
            h = self._host[slot][key] = torch.empty(shape, dtype=torch.uint8).pin_memory()
        torch.index_select(src, 0, torch.from_numpy(ids), out=h)
        return h, self._dev_out(slot, key, shape)

    def prefetch(self, idx, draws, aug=None):
        s = self.index.sample(idx, draws)
        n, T = s["frames"].shape
        check_ids(np.concatenate([s["frames"].reshape(-1), s["goal"]]), min(int(v.shape[0]) for v in self.frames.values()))
        slot = self._slot
        self._slot ^= 1
        # Ring ordering.  The slot's device buffers were handed out two batches ago; the step that reads them has been
        # enqueued on the compute stream by now (prefetch is called after it), so the copy stream waits for everything
        # enqueued there so far - not for the step that consumes the OTHER slot next, which it overlaps.
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))
        # ... and the host may not rewrite a pinned staging buffer while the copy issued from it is still in flight
        if self._copied[slot] is not None:
            self._copied[slot].synchronize()
        jobs = [("states", c, (n, T), ("s", c), fr, s["frames"].reshape(-1)) for c, fr in self.frames.items()]
        jobs += [("goal", c, (n,), ("g", c), fr, s["goal"]) for c, fr in self.frames.items()]
        acts = torch.from_numpy(pad_actions(self.actions, s["frames"], s["padded"])).pin_memory()
        b = {"states": {}, "goal": {}}
        if self.gather == "host":
            staged = [self._stage(slot, key, fr, ids) for _, _, _, key, fr, ids in jobs]
        with torch.cuda.stream(self.stream):
            for j, (kind, c, lead, key, fr, ids) in enumerate(jobs):
                if self.gather == "host":
                    h, d = staged[j]
                    d.copy_(h, non_blocking=True)
                else:
                    d = self._dev_out(slot, key, (len(ids),) + tuple(fr.shape[1:]))
                    ids_d = torch.from_numpy(np.ascontiguousarray(ids, dtype=np.int64)).pin_memory().to(self.dev, non_blocking=True)
                    ops.gather_frames_u8(fr, ids_d, d)  # fr is mapped host memory: the reads cross PCIe
                b[kind][c] = d.view(*lead, *d.shape[1:])
            b["actions"] = acts.to(self.dev, non_blocking=True)
            b["disp"] = torch.from_numpy(s["disp"]).pin_memory().to(self.dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._copied[slot] = ev
        b.update(idx=torch.from_numpy(s["idx"]), window_size=torch.from_numpy(s["window_size"]))
        if aug is not None:
            b["aug"] = aug
        self._pending = (b, ev)

    def next(self):
        """The prefetched batch; the compute stream waits for its copy."""
        b, ev = self._pending
        cs = torch.cuda.current_stream(self.dev)
        cs.wait_event(ev)
        # tensors allocated under the copy stream are read on the compute stream: tell the caching allocator
        for t in (b["actions"], b["disp"], *b["states"].values(), *b["goal"].values()):
            t.record_stream(cs)
        self._pending = None
        return b
