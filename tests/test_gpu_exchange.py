"""The library-side step exchange (csrc/agx_exchange.hip) on one GPU: a world of one still goes
through ncclCommInitRank / ncclAllGather, the worker thread, both streams and all four events.
World size 2 on the one-GPU box: two processes share the device and the exchange binds a test double of the five
RCCL entry points (tests/fakerccl: a stream-ordered shared-memory all-gather; RCCL refuses two ranks on one device),
see test_rccl_thread_exchange_world2."""
import os
import socket
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture()
def world_of_one():
    import torch.distributed as dist

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("library_backend", ["rccl_thread", "peer_push"])
def test_library_exchange_matches_process_group(world_of_one, library_backend):
    from aerial_gym_simulator_amd.sharding import StepGather

    dev = torch.device("cuda:0")
    n, d = 1000, 13
    a = StepGather(n, d, dev, backend=library_backend)
    b = StepGather(n, d, dev, backend="process_group")
    assert a.backend == library_backend and b.backend == "process_group"
    if library_backend == "peer_push":
        assert a.push_selftest == "passed"  # a word and a flag went through every mapping before the first post
    assert StepGather(n, d, dev).backend == "peer_push"  # what "auto" picks on a HIP device (rccl_thread if it cannot be set up)
    g = torch.Generator(device=dev).manual_seed(3)
    # synchronous form: this step's rows
    for step in range(6):
        p = step & 1
        rows = torch.rand(n, d + 3, device=dev, generator=g)
        for x in (a, b):
            x.rows[p].copy_(rows)
        ga, gb = a.exchange(p, overlap=False), b.exchange(p, overlap=False)
        torch.cuda.synchronize()
        assert torch.equal(ga, rows) and torch.equal(gb, rows)
    # overlapped form: the previous step's rows come back, nothing on the first call
    a._last = b._last = None
    sent = []
    for step in range(40):
        p = step & 1
        rows = torch.rand(n, d + 3, device=dev, generator=g)
        sent.append(rows)
        for x in (a, b):
            x.rows[p].copy_(rows)
        ga, gb = a.exchange(p, overlap=True), b.exchange(p, overlap=True)
        if step == 0:
            assert ga is None and gb is None
        else:
            ga, gb = ga.clone(), gb.clone()  # stream-ordered snapshot before the next overwrite
            torch.cuda.synchronize()
            assert torch.equal(ga, sent[step - 1]) and torch.equal(gb, sent[step - 1]), step
        if step == 20:
            time.sleep(0.05)  # let the worker thread fall asleep: the next post must wake it
    a.flush()
    b.flush()
    torch.cuda.synchronize()
    last_a = a.gathered[a._slot_of_parity[1]] if a.backend == "peer_push" else a.gathered[1]  # (peer push: four receive slots)
    assert torch.equal(last_a, sent[-1]) and torch.equal(b.gathered[1], sent[-1])
    a.close()
    a.close()  # idempotent


def test_auto_falls_back_to_the_rccl_thread_when_peer_push_cannot_be_set_up(world_of_one, monkeypatch):
    """A platform that refuses the peer mappings (injected with a test double: the library object's agx_exchange_create_push is
    replaced by a function that fails -- the product has no hook for it): every rank learns it before anything is built
    (StepGather._agree), "auto" takes the RCCL worker thread instead, an explicit backend="peer_push" raises."""
    from aerial_gym_simulator_amd import _lib
    from aerial_gym_simulator_amd.sharding import StepGather

    dev = torch.device("cuda:0")
    monkeypatch.setattr(_lib.load(), "agx_exchange_create_push", lambda *a: 1)
    sg = StepGather(256, 13, dev)
    assert sg.backend == "rccl_thread"
    rows = torch.rand(256, 16, device=dev)
    sg.rows[0].copy_(rows)
    got = sg.exchange(0, overlap=False).clone()
    torch.cuda.synchronize()
    assert torch.equal(got, rows)
    sg.close()
    with pytest.raises(RuntimeError, match="peer_push exchange unavailable"):
        StepGather(256, 13, dev, backend="peer_push")


def test_worker_thread_push_whatever_queue_its_stream_lands_on(world_of_one):
    """HIP maps streams onto a few hardware queues in creation order.  The consumer of the worker-thread push must not spin
    on this rank's own arrival flag: if its stream shares a queue with the communication stream the spinning kernel sits in
    front of the push kernel it waits for (found as an order-dependent 10 s time-out).  Shift the assignment by creating
    0 .. 7 streams first; every variant must exchange promptly."""
    from aerial_gym_simulator_amd.sharding import StepGather

    dev = torch.device("cuda:0")
    n, d = 512, 13
    keep = []
    g = torch.Generator(device=dev).manual_seed(5)
    for extra in range(8):
        keep.append(torch.cuda.Stream(device=dev))
        x = StepGather(n, d, dev, backend="peer_push")
        for consumer in (torch.cuda.current_stream(dev), keep[-1]):
            with torch.cuda.stream(consumer):
                for step in range(4):
                    rows = torch.rand(n, d + 3, device=dev, generator=g)
                    consumer.wait_stream(torch.cuda.current_stream(dev))
                    x.rows[step & 1].copy_(rows)
                    t0 = time.time()
                    got = x.exchange(step & 1, overlap=False).clone()
                    torch.cuda.synchronize()
                    assert time.time() - t0 < 2.0, (extra, step)
                    assert torch.equal(got, rows), (extra, step)
        x.close()


@pytest.mark.parametrize("library_backend", ["rccl_thread", "peer_push"])
def test_library_exchange_orders_against_the_stepping_stream(world_of_one, library_backend):
    """2000 back-to-back steps on a side stream: every gathered buffer must hold exactly the rows of its step
    (the rows are rewritten by the very next-but-one step, so a missing stream dependency shows up as a
    newer or older counter)."""
    from aerial_gym_simulator_amd.sharding import StepGather

    dev = torch.device("cuda:0")
    n, d = 8192, 13
    x = StepGather(n, d, dev, backend=library_backend)
    side = torch.cuda.Stream(device=dev)
    seen = torch.zeros(2000, device=dev)
    with torch.cuda.stream(side):
        for step in range(2000):
            p = step & 1
            x.rows[p].fill_(float(step))  # stands in for the observation kernel of this step
            buf = x.exchange(p, overlap=True)
            if buf is not None:
                seen[step - 1] = buf[::97].max() + buf[::89].min()  # both == step - 1
        x.flush()
        last = (x.gathered[x._slot_of_parity[1]] if x.backend == "peer_push" else x.gathered[1]).clone()
    side.synchronize()
    assert torch.equal(seen[:-1].cpu(), 2.0 * torch.arange(1999, dtype=torch.float32))
    assert float(last.min()) == float(last.max()) == 1999.0
    x.close()


def test_exchange_argument_errors(world_of_one):
    import ctypes as C

    from aerial_gym_simulator_amd import _lib
    from aerial_gym_simulator_amd.sharding import StepGather

    lib = _lib.load()
    assert lib.agx_exchange_unique_id(None, None, 128) != 0
    uid = (C.c_char * 128)()
    assert lib.agx_exchange_unique_id(b"/nonexistent/librccl.so", uid, 64) != 0
    h = C.c_void_p()
    assert lib.agx_exchange_create(None, uid.raw, 128, 3, 2, 0, C.byref(h)) != 0  # rank outside the world
    assert "rank" in lib.agx_last_error().decode()
    with pytest.raises(ValueError):
        StepGather(4, 13, torch.device("cuda:0"), backend="mpi")
    assert lib.agx_exchange_destroy(None) == 0


@pytest.mark.parametrize("kernel_push", [None, True])
@pytest.mark.parametrize("which", ["position", "navigation"])
@pytest.mark.parametrize("ready", ["signal", "event"])
@pytest.mark.parametrize("library_backend", ["rccl_thread", "peer_push"])
def test_library_exchange_of_a_stepping_task(world_of_one, which, ready, library_backend, kernel_push):
    """The rows the observation kernels write travel through the library-side exchange, ordered by the
    kernels' own step_signal flag (or by an event): every gathered buffer is exactly the step's
    obs | reward | terminated | truncated, in the synchronous and in the overlapped form, across resets."""
    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.config.task_config import navigation_task_config, position_setpoint_task_config
    from aerial_gym_simulator_amd.registry.task_registry import task_registry
    from aerial_gym_simulator_amd.sharding import StepGather

    if kernel_push and not (library_backend == "peer_push" and which == "navigation"):
        pytest.skip("kernel_push=True differs from the default only for the wide rows of the sensor tasks under peer push")
    dev = "cuda:0"
    cfg = position_setpoint_task_config if which == "position" else navigation_task_config
    old = (cfg.episode_len_steps, cfg.args, getattr(cfg, "controller_name", None))
    cfg.device, cfg.episode_len_steps, cfg.args = dev, 9, {}
    if which == "position":
        cfg.controller_name = "lee_position_control"
    try:
        n = 8192 if which == "position" else 96
        task = task_registry.make_task(which + ("_setpoint_task" if which == "position" else "_task"), seed=5, num_envs=n, headless=True)
        task.reset()
        d = task.task_obs["observations"].shape[1]
        sg = StepGather(n, d, dev, env=task.sim_env, reward=task.rewards, backend=library_backend, ready=ready, kernel_push=kernel_push)
        assert ready == "signal" or sg.signal is None or sg._kernel_push  # "signal" may fall back to events (agx_exchange_probe)
        g = torch.Generator(device=dev).manual_seed(9)
        acts = [torch.rand(n, 4, device=dev, generator=g) * 2 - 1 for _ in range(4)]

        def snapshot(ret):
            obs, rew, term, trunc, _ = ret
            return torch.cat([obs["observations"], rew[:, None], term.float()[:, None], trunc.float()[:, None]], dim=1)

        steps = 60 if which == "position" else 12
        for step in range(steps):  # synchronous form
            want = snapshot(task.step(acts[step & 3]))
            got = sg.exchange(task.sim_env._parity, overlap=False).clone()
            assert torch.equal(got, want), (which, ready, step)
        sg.flush()
        sg._last = None
        hist, checks, truncs = [], [], 0  # (sg.lag: the overlapped form returns the rows of 1 step ago, 2 when the kernels push them)
        for step in range(steps):  # overlapped form, no host synchronisation inside the loop
            ret = task.step(acts[step & 3])
            hist.append(snapshot(ret))
            truncs += int(ret[3].sum()) if step % 5 == 0 else 0
            buf = sg.exchange(task.sim_env._parity, overlap=True)
            if len(hist) > sg.lag:
                checks.append((buf.clone(), hist[-1 - sg.lag]))
        sg.flush()
        torch.cuda.synchronize()
        assert len(checks) == steps - sg.lag
        for step, (got, want) in enumerate(checks):
            assert torch.equal(got, want), (which, ready, step)
        if which == "position":
            # long run without any host synchronisation: a row read before it was visible device-wide (the flag
            # protocol of agx_step_signal.h) would show up as a mismatch counted on the device
            bad = torch.zeros((), device=dev, dtype=torch.int64)
            hist = []
            sg.flush()
            sg._last = None
            for step in range(2000):
                hist.append(snapshot(task.step(acts[step & 3])))
                buf = sg.exchange(task.sim_env._parity, overlap=True)
                if len(hist) > sg.lag:
                    bad += (buf != hist[-1 - sg.lag]).sum()
                hist = hist[-3:]
            sg.flush()
            torch.cuda.synchronize()
            assert int(bad) == 0
        if sg.signal is not None and not sg._kernel_push:  # (rows pushed by the kernels: flags are raised by the next step's first kernel)
            assert int(sg.signal[2]) == 0  # the arrival counter is back at zero after every launch
            assert int(sg.signal[:2].max()) == task.sim_env.step_counter
        # the observation kernels push 16-float rows themselves; wide rows go through the library's copy kernel unless asked otherwise
        assert sg._kernel_push == (library_backend == "peer_push" and (which == "position" or bool(kernel_push)))
        sg.close()
        assert task.sim_env._buffers.step_signal is None and task.sim_env._buffers.push_world == 0
        task.step(acts[0])  # stepping goes on without the exchange
        torch.cuda.synchronize()
    finally:
        cfg.episode_len_steps, cfg.args = old[0], old[1]
        if old[2] is not None:
            cfg.controller_name = old[2]
