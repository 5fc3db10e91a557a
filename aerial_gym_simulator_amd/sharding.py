"""Multi-GPU env sharding: one process per GPU, envs split contiguously, ONE collective per
env step (RCCL all-gather over xGMI of the packed observation | reward | termination |
truncation rows).  The simulator itself needs no communication: envs never read each other
(SURVEY.md section 8e); the reference has no distributed path at all.

Payload per rank and step: N_local * (obs_dim + 3) * 4 B  (8192 envs, 13-D obs: 0.5 MB), i.e.
latency bound on a fully connected xGMI node -- hence
  * a single fused buffer, not one collective per tensor;
  * the send rows are written by the observation kernel itself (AgxEnvBuffers.step_rows), so
    the exchange adds no launch to the step;
  * rows and receive buffers are double buffered by step parity and the collective runs
    asynchronously on its own stream: the gather of step t overlaps the kernels of step t+1;
  * on HIP devices the collective is enqueued by a worker thread of the simulator library on a
    communicator of its own (csrc/agx_exchange.hip, `backend="rccl_thread"`): the stepping thread
    pays one event record + one stream wait per step.  Going through torch's process group
    (`backend="process_group"`, the only choice on CPU/gloo) costs 27 us of host time per step --
    more than the 18 us dynamics-only step itself (profiles/r01_exchange_probe.txt);
  * `backend="peer_push"` (round 3, what "auto" picks on HIP devices) runs NO collective kernel per step: every rank
    stores its rows straight into every peer's receive buffer (mapped through hipIpcMemHandle, one xGMI link per
    destination) and raises an arrival flag there; the consumer's stream waits on its own flags.  An RCCL all-gather
    costs 12.5 us per launch even in a world of one -- as long as the dynamics-only step -- and consecutive gathers
    serialise on the communication stream; the push kernel moves 0.5 MB per destination and nothing else.  RCCL /
    torch.distributed carry only the set-up (the 128 handle bytes per rank).
"""
import ctypes as C
import os

import torch
import torch.distributed as dist


def rccl_library_path():
    """The librccl.so this process already uses (torch bundles its own); None = loader default.  (Another library is bound
    only through the explicit `StepGather(rccl_library=...)` argument: the world-size-2 tests on a one-GPU box hand in a test
    double, because RCCL refuses two ranks on one device.  No environment variable reaches this choice.)"""
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return cand if os.path.exists(cand) else None


def shard_range(total_envs, rank, world_size):
    """Contiguous [lo, hi) env range of `rank`; remainders go to the first ranks."""
    base, rem = divmod(total_envs, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def semantic_id_offset(rank, envs_per_rank, assets_per_env):
    """Start of this rank's slice of the reference's global segmentation counter
    (env_manager.py:147: 100 + one id per asset, counted across all envs)."""
    return rank * envs_per_rank * assets_per_env


_PARKED = []  # exchanges abandoned without a collective close(): kept mapped for the life of the process (StepGather.close)


class StepGather:
    """All-gathers [N_local, obs_dim + 3] fp32 rows (obs | reward | terminated | truncated) once
    per env step.

    With `env` (an EnvManager on a HIP device) the rows are produced by the observation kernels;
    without, the caller fills them with `pack()` (host-logic tests, custom tasks).

    exchange(parity, overlap) returns the gathered [world * N_local, obs_dim + 3] buffer (no
    device work besides the collective; `unpack` slices it):
      overlap=False  THIS step's rows of all ranks (stream-ordered);
      overlap=True   this step's collective is left in flight and the buffer of the PREVIOUS
                     step is returned (None on the first call) -- asynchronous samplers such
                     as the reference's Sample Factory recipe tolerate the one-step delay.
    """

    KERNEL_PUSH_MAX_ROW = 16  # floats: rows up to this size are stored at every destination by the observation kernel itself

    def __init__(self, num_envs_local, obs_dim, device, env=None, reward=None, group=None, backend="auto", ready="signal",
                 kernel_push=None, rccl_library=None, push_selftest=True):
        """backend: "process_group" (torch.distributed collective), "rccl_thread" (the library's worker
        thread on its own RCCL communicator; HIP devices only) or "auto" (rccl_thread when the group runs
        on RCCL and every rank could set it up, else process_group).
        ready (rccl_thread with `env` only): "signal" = the observation kernel publishes a flag in device
        memory that the communication stream spins on (no HIP call per step on this thread); "event" = an
        event recorded after the step (what rows filled by `pack()` always use).
        kernel_push (peer_push with `env`): True = the observation kernels store the rows into every rank's receive buffer
        themselves; False = they store them locally and the library's push kernel copies them (16-byte coalesced) behind the
        step's flag; None = by row size: the 16-float row of the position task is written by its kernels as 16-byte quarters
        (one 64-byte line per env and destination), and a microsecond-scale step cannot afford a second host thread
        launching kernels; the wide rows of the sensor tasks (84 / 340 floats) are produced element by element -- 4-byte
        stores across xGMI -- so they go through the copy kernel, whose launch is nothing next to a millisecond step."""
        self.group = group
        self._want_push_selftest = bool(push_selftest)  # peer_push: a word and a flag through every mapping before the first post
        self._rccl_library = rccl_library  # tests only: path of the collective library `rccl_thread` binds (default: torch's RCCL)
        self.collective = dist.is_initialized()  # a world of one still goes through RCCL (bench debugging aid)
        self.world = dist.get_world_size(group) if self.collective else 1
        self.n, self.obs_dim = num_envs_local, obs_dim
        self.device = torch.device(device)
        self.rows = torch.zeros(2, num_envs_local, obs_dim + 3, device=device)
        self.gathered = torch.zeros(2, self.world * num_envs_local, obs_dim + 3, device=device) if self.collective else self.rows
        self._work = [None, None]
        self._last = None
        self._native = None
        self._env = env
        self.signal = None
        if ready not in ("signal", "event"):
            raise ValueError(f"unknown ready mode {ready!r}")
        if backend not in ("auto", "process_group", "rccl_thread", "peer_push"):
            raise ValueError(f"unknown exchange backend {backend!r}")
        if backend in ("rccl_thread", "peer_push") and not (self.collective and self.device.type == "cuda"):
            raise RuntimeError(f"backend={backend!r} needs an initialised process group and a HIP device")
        self._push = False
        self._posts = 0
        self._slot_of_parity = [0, 0]
        if self.collective and self.device.type == "cuda" and backend in ("auto", "peer_push") and (
                backend == "peer_push" or "nccl" in str(dist.get_backend(group))):
            self._native = self._create_push(required=backend == "peer_push")
            self._push = self._native is not None
        if self._native is None and self.collective and self.device.type == "cuda" and backend in ("auto", "rccl_thread") and (
                backend == "rccl_thread" or "nccl" in str(dist.get_backend(group))):  # "auto" never picks it over gloo
            self._native = self._create_native(required=backend == "rccl_thread")
        self.backend = ("peer_push" if self._push else "rccl_thread") if self._native is not None else ("process_group" if self.collective else "none")
        self.lag = 1            # exchange(overlap=True) returns the rows of `lag` steps ago
        self._kernel_push = False
        kernel_push_forced = kernel_push is True
        if kernel_push is None:
            kernel_push = obs_dim + 3 <= self.KERNEL_PUSH_MAX_ROW
        # kernel push covers one node (8 ranks: AgxEnvBuffers.push_delta[7]) and keeps slot / sequence number in host-side kernel
        # arguments, which a captured step graph would freeze: both cases go through the copy-kernel push instead
        if kernel_push and (self.world > 8 or bool(getattr(env, "step_graph_mode", False))):
            if kernel_push_forced:
                raise ValueError("kernel_push=True needs world <= 8 and an eagerly stepped task (not hipGraph replay)")
            kernel_push = False
        if env is not None and self._push and kernel_push and not getattr(env, "rows_written_twice_per_step", False):
            # the observation kernels store their rows at every destination themselves: nothing per step on the host, no
            # launch, no worker thread; the buffer of TWO steps ago is complete by construction when a step's kernels ran
            recv, flags, tout = (C.c_void_p * self.world)(), (C.c_void_p * self.world)(), C.c_void_p()
            from . import _lib

            _lib.check(self._lib.agx_exchange_push_peers(self._native, recv, flags, C.byref(tout)), "agx_exchange_push_peers")
            self.signal = torch.zeros(4, dtype=torch.int32, device=device)
            env.bind_peer_push([int(x) for x in recv], [int(x) for x in flags], dist.get_rank(self.group), self.world, self.push_slots,
                               self.obs_dim + 3, reward, self.signal, tout.value)
            self._kernel_push, self.lag = True, 2
            return
        if env is not None:
            if getattr(env, "rows_written_twice_per_step", False):
                # return_state_before_reset: the observation kernel runs before AND after the reset, each publishing the
                # step's flag -- the gather could start between the two and read torn rows: order by an event instead
                ready = "event"
            if self._native is not None and ready == "signal":
                rc = self._lib.agx_exchange_probe(self._native, torch.cuda.current_stream(self.device).cuda_stream)
                if rc < 0:
                    from . import _lib

                    _lib.check(rc, "agx_exchange_probe")
                if rc == 1:  # else: no communication stream independent of this one -> events
                    self.signal = torch.zeros(4, dtype=torch.int32, device=device)
                    self._signal_ptr = self.signal.data_ptr()
            env.bind_step_rows(self.rows, reward, self.signal)

    def _create_native(self, required):
        from . import _lib

        lib = _lib.load()
        path = self._rccl_library or rccl_library_path()
        cpath = path.encode() if path else None
        uid = (C.c_char * 128)()
        # every rank binds RCCL and draws an id (only rank 0's is used): a rank that cannot must be known
        # BEFORE ncclCommInitRank, which blocks until all ranks have joined
        rc = lib.agx_exchange_unique_id(cpath, uid, 128)
        err = lib.agx_last_error().decode("utf-8", "replace") if rc else ""
        # control traffic (one flag, the 128 id bytes) travels on whatever the process group runs on
        ctl = self.device if "nccl" in str(dist.get_backend(self.group)) else torch.device("cpu")
        ok = torch.tensor([1 if rc == 0 else 0], device=ctl, dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        if int(ok.item()) == 0:
            if required:
                raise RuntimeError(f"rccl_thread exchange unavailable on some rank ({err or 'see the other ranks'})")
            return None
        idt = torch.tensor(list(uid.raw), dtype=torch.uint8, device=ctl)
        dist.broadcast(idt, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        if ctl.type == "cuda":
            torch.cuda.synchronize(self.device)
        raw = bytes(idt.cpu().tolist())
        handle = C.c_void_p()
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _lib.check(lib.agx_exchange_create(cpath, raw, 128, dist.get_rank(self.group), self.world, index, C.byref(handle)),
                   "agx_exchange_create")
        self._lib = lib
        self._count = self.n * (self.obs_dim + 3)
        self._ptr = [(self.rows[p].data_ptr(), self.gathered[p].data_ptr()) for p in (0, 1)]  # tensor indexing costs microseconds
        return handle

    def _agree(self, ok):
        """all ranks learn whether every rank got this far (control traffic on whatever the process group runs on)"""
        ctl = self.device if "nccl" in str(dist.get_backend(self.group)) else torch.device("cpu")
        t = torch.tensor([1 if ok else 0], device=ctl, dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return int(t.item()) == 1, ctl

    PUSH_SELFTEST_MS = 5000
    push_selftest = None

    def _create_push(self, required):
        """peer push: allocate this rank's receive buffer, trade the IPC handles, map the peers' buffers"""
        from . import _lib

        lib = _lib.load()
        self._lib = lib
        self._count = self.n * (self.obs_dim + 3)
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        handle = C.c_void_p()
        rc = lib.agx_exchange_create_push(dist.get_rank(self.group), self.world, index, self._count, C.byref(handle))
        err = lib.agx_last_error().decode("utf-8", "replace") if rc else ""
        mine = (C.c_char * 128)()
        if rc == 0:
            rc = lib.agx_exchange_push_export(handle, mine, 128)
            err = lib.agx_last_error().decode("utf-8", "replace") if rc else ""
        ok, ctl = self._agree(rc == 0)
        if ok:
            all_h = torch.zeros(self.world * 128, dtype=torch.uint8, device=ctl)
            dist.all_gather_into_tensor(all_h, torch.tensor(list(mine.raw), dtype=torch.uint8, device=ctl), group=self.group)
            if ctl.type == "cuda":
                torch.cuda.synchronize(self.device)
            raw = bytes(all_h.cpu().tolist())
            rc = lib.agx_exchange_push_connect(handle, raw, len(raw))
            err = lib.agx_last_error().decode("utf-8", "replace") if rc else ""
            ok, _ = self._agree(rc == 0)
            if ok and self._want_push_selftest:  # mapped everywhere: do stores through the mappings arrive?  (one word and one flag per pair of ranks)
                passed = C.c_int(0)
                rc = lib.agx_exchange_push_selftest(handle, self.PUSH_SELFTEST_MS, C.byref(passed),
                                                    torch.cuda.current_stream(self.device).cuda_stream)
                if rc or not passed.value:
                    err = lib.agx_last_error().decode("utf-8", "replace")
                ok, _ = self._agree(rc == 0 and passed.value == 1)
                self.push_selftest = "passed" if ok else f"failed ({err or 'on another rank'})"
        if not ok:
            if handle.value:
                lib.agx_exchange_destroy(handle)
            if required:
                raise RuntimeError(f"peer_push exchange unavailable on some rank ({err or 'see the other ranks'})")
            return None
        recv, slots, unc = C.c_void_p(), C.c_int(), C.c_int()
        _lib.check(lib.agx_exchange_push_buffer(handle, C.byref(recv), C.byref(slots), C.byref(unc)), "agx_exchange_push_buffer")
        self.push_slots, self.push_flags_uncached = slots.value, bool(unc.value)

        class _Mem:  # the library's allocation as a tensor (no copy): [slots, world * N, obs_dim + 3]
            __cuda_array_interface__ = {"shape": (slots.value, self.world * self.n, self.obs_dim + 3), "typestr": "<f4",
                                        "data": (recv.value, False), "version": 2, "strides": None}

        self._push_mem = _Mem()
        self.gathered = torch.as_tensor(self._push_mem, device=self.device)
        self._ptr = [(self.rows[p].data_ptr(), None) for p in (0, 1)]
        dist.barrier(group=self.group)  # nobody posts before every rank has mapped every buffer
        return handle

    def comm_info(self):
        """(rank, world size) as the exchange's communicator reports them (ncclCommUserRank / ncclCommCount) for
        `rccl_thread`, the process group's for `process_group`, (0, 1) without a collective."""
        if self._native is not None:
            from . import _lib

            r, w = C.c_int(-1), C.c_int(-1)
            _lib.check(self._lib.agx_exchange_info(self._native, C.byref(r), C.byref(w)), "agx_exchange_info")
            return r.value, w.value
        if self.collective:
            return dist.get_rank(self.group), dist.get_world_size(self.group)
        return 0, 1

    def close(self, collective=True):
        """Drains and frees the library-side exchange.  With peer push an explicit close() is REQUIRED ON EVERY RANK and is
        collective (a barrier keeps any rank from unmapping a buffer a peer's kernels may still be storing into).
        `collective=False` -- what the finalizer uses -- skips the barrier: a rank that is torn down alone (garbage collection,
        an exception path) must never enter a collective its peers will not join; with no barrier nobody knows whether a peer
        is still storing into this rank's receive buffer, so that path only stops THIS rank's kernels from pushing and parks
        the library handle (receive buffer and mappings stay alive until the process exits) instead of freeing it.  After
        close() the gathered buffer is gone (`self.gathered` is None: every slice exchange() handed out aliased the library's
        allocation -- copy what you keep)."""
        h, self._native = self._native, None
        if h is not None and self._push and not collective:
            if self._kernel_push:
                torch.cuda.synchronize(self.device)
                self._env.unbind_peer_push()
                self._kernel_push = False
            if self.signal is not None and self._env is not None:
                # an env that keeps stepping after its exchange was dropped must not keep signalling the parked handle every step
                self._env.bind_step_rows(self.rows, self._env._step_rows[1], None)
                self.signal = None
            self.gathered, self._push_mem = None, None
            _PARKED.append((self._lib, h))  # a peer's store_peer / store_peer4 may still be in flight: never unmapped from here
            return
        if h is not None and self._kernel_push:
            torch.cuda.synchronize(self.device)
            if collective and dist.is_initialized():
                try:
                    dist.barrier(group=self.group)  # no rank unmaps a buffer a peer's kernels may still be storing into
                except Exception:  # noqa: BLE001  (a peer that is already gone)
                    pass
            self._env.unbind_peer_push()
            self._kernel_push = False
        if h is not None:
            if self._push:
                self.gathered = None  # views of the library's receive buffer (freed by the next line) must fail loudly, not read freed memory
                self._push_mem = None
            self._lib.agx_exchange_destroy(h)  # drains the communication stream
            if self.signal is not None and self._env is not None:
                self._env.bind_step_rows(self.rows, self._env._step_rows[1], None)
                self.signal = None

    def __del__(self):
        try:
            self.close(collective=False)
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    def pack(self, parity, obs, reward, terminated, truncated):
        p, d = self.rows[parity], self.obs_dim
        p[:, :d] = obs
        p[:, d] = reward
        p[:, d + 1] = terminated
        p[:, d + 2] = truncated
        return p

    def exchange(self, parity, overlap=False):
        if self._kernel_push:
            return self._exchange_kernel_push(parity, overlap)
        if self._native is not None:
            return self._exchange_native(parity, overlap)
        if self.collective:
            self._work[parity] = dist.all_gather_into_tensor(self.gathered[parity], self.rows[parity], group=self.group,
                                                             async_op=True)
        if not overlap:
            self.wait(parity)
            return self.gathered[parity]
        prev, self._last = self._last, parity
        if prev is None:
            return None
        # the next step writes rows[prev] again: its collective must have drained (stream wait, no host block)
        self.wait(prev)
        return self.gathered[prev]

    def _exchange_kernel_push(self, parity, overlap):
        """rows pushed by the observation kernels (env.bind_peer_push): `overlap` returns the rows of TWO steps ago -- complete
        once the stream has run this step's kernels, which waited for exactly that -- at no cost; the synchronous form puts a
        one-wave flag wait for THIS step's rows on the stream"""
        seq = self._env._buffers.push_seq
        if (seq & 255) == 0:  # a device-side wait that gave up (10 s) says so through a host-visible word
            from . import _lib

            _lib.check(self._lib.agx_exchange_check(self._native), "agx_exchange_check")
        if not overlap:
            if seq == 0:
                return None  # nothing has been pushed yet (no step ran since the exchange was bound)
            from . import _lib

            _lib.check(self._lib.agx_exchange_push_wait_seq(self._native, seq & 0xFFFFFFFF, self._env._stream()), "agx_exchange_push_wait_seq")
            return self.gathered[(seq - 1) % self.push_slots]
        return self.gathered[(seq - 3) % self.push_slots] if seq > 2 else None

    def _exchange_native(self, parity, overlap):
        from . import _lib

        # the stream the step just used (looked up once per step by the env); torch's lookup costs a microsecond
        stream = self._env._stream() if self._env is not None else torch.cuda.current_stream(self.device).cuda_stream
        if overlap:
            prev, self._last = self._last, parity
            wait_parity = -1 if prev is None else prev
        else:
            prev = wait_parity = parity
        send, recv = self._ptr[parity]
        if self._push:  # the rows gathered by the s-th post lie in slot (s - 1) % slots of the exchange's own buffer
            self._posts += 1
            new_slot = (self._posts - 1) % self.push_slots
            prev_slot = new_slot if not overlap else (self._slot_of_parity[prev] if prev is not None else None)
            self._slot_of_parity[parity] = new_slot
        if self.signal is not None:  # the kernels of the step just enqueued publish signal[parity] = step_counter
            rc = self._lib.agx_exchange_step(self._native, parity, send, recv, self._count, self._signal_ptr,
                                             self._env.step_counter & 0x7FFFFFFF, wait_parity, stream)
        else:
            rc = self._lib.agx_exchange_step(self._native, parity, send, recv, self._count, None, 0, wait_parity, stream)
        if rc:
            _lib.check(rc, "agx_exchange_step")
        if prev is None:
            return None
        return self.gathered[prev_slot] if self._push else self.gathered[prev]

    def wait(self, parity):
        if self._kernel_push:  # everything pushed so far has arrived (parity has no meaning here: the slots are sequence numbers)
            seq = self._env._buffers.push_seq
            if seq > 0:
                from . import _lib

                _lib.check(self._lib.agx_exchange_push_wait_seq(self._native, seq & 0xFFFFFFFF, torch.cuda.current_stream(self.device).cuda_stream),
                           "agx_exchange_push_wait_seq")
            return
        if self._native is not None:
            from . import _lib

            _lib.check(self._lib.agx_exchange_wait(self._native, parity, torch.cuda.current_stream(self.device).cuda_stream),
                       "agx_exchange_wait")
            return
        w, self._work[parity] = self._work[parity], None
        if w is not None:
            w.wait()

    def flush(self):
        self.wait(0)
        self.wait(1)

    def unpack(self, buf):
        """(obs, reward, terminated, truncated) of a buffer returned by exchange()."""
        d = self.obs_dim
        return buf[:, :d], buf[:, d], buf[:, d + 1] > 0.5, buf[:, d + 2] > 0.5
