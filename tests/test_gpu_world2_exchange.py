"""GPU: the library-side step exchange (csrc/agx_exchange.hip through sharding.StepGather, backend "rccl_thread") at WORLD
SIZE 2 on a one-GPU box: two processes, a test double for RCCL (tests/fakerccl) that gathers through shared memory.  These
tests run last of the GPU tests (file name).  A run whose two processes do not finish in time FAILS, with what each rank
said: a hang is exactly what these tests exist to catch.  Only on a box known to be loaded may it be repeated, by opting
in with AGX_WORLD2_RETRIES=<n> (two processes rendezvousing through host functions on one shared GPU is the one place
where the test infrastructure itself can stall); a wrong result fails at once either way."""
import os
import socket
import time

import pytest

pytestmark = pytest.mark.gpu


def _run_world2(mode, steps, extra_env=None, timeout=420):
    attempts = 1 + max(0, int(os.environ.get("AGX_WORLD2_RETRIES", "0")))
    for attempt in range(attempts):
        try:
            return _run_world2_once(mode, steps, extra_env, timeout)
        except TimeoutError as e:
            if attempt + 1 == attempts:
                raise AssertionError(str(e))
            print(f"[world-2 exchange] attempt {attempt + 1} did not finish, repeating (AGX_WORLD2_RETRIES):\n{e}", flush=True)


def _rendezvous_port():
    """a port BELOW the kernel's ephemeral range (32768+): `bind(0)` hands out an ephemeral port that an outgoing connection of
    any process may take again between our close() and rank 0's listen() -- seen once in round 4 as EADDRINUSE in rank 0 and a
    420-s wait for rank 1"""
    import random

    rng = random.Random(os.getpid() ^ int(time.time() * 1000))
    for _ in range(200):
        port = rng.randrange(15000, 30000)
        with socket.socket() as s:
            try:
                s.bind(("127.0.0.1", port))
            except OSError:
                continue
            return port
    raise RuntimeError("no free rendezvous port")


def _run_world2_once(mode, steps, extra_env, timeout):
    import subprocess
    import sys

    import importlib.util

    spec = importlib.util.spec_from_file_location(
        "agx_fakerccl_build", os.path.join(os.path.dirname(os.path.abspath(__file__)), "fakerccl", "build.py"))
    fake_build = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fake_build)
    lib = fake_build.build()
    port = _rendezvous_port()
    # (the double's rendezvous bound is generous here: on a loaded box one rank's set-up can trail the other's by tens of
    #  seconds; the failure-path test sets its own, short one)
    env = dict(os.environ, AGX_TEST_FAKERCCL=lib, HSA_ENABLE_IPC_MODE_LEGACY="0", AGX_FAKERCCL_TIMEOUT_S="150")
    env.update(extra_env or {})
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "exchange_world2_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", str(port), mode, str(steps)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [None] * len(procs)
    t0 = time.time()
    dead_since = None
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        # a rank that died on its own leaves its peer waiting inside a rendezvous or a collective: give the peer a moment to
        # notice (the library's own bounds are shorter than this harness's), then end the run with what both said
        if any(c not in (None, 0) for c in codes):
            dead_since = dead_since or time.time()
        expired = time.time() - t0 > timeout or (dead_since is not None and time.time() - dead_since > 60)
        if expired:
            for q in procs:
                if q.poll() is None:
                    q.kill()
            tails = []
            for r, q in enumerate(procs):  # what each rank printed before it was killed (a rank that died leaves its peer waiting)
                try:
                    tails.append(f"--- rank {r} (exit code {codes[r]}):\n" + (q.communicate(timeout=10)[0] or "")[-2500:])
                except Exception as e:  # noqa: BLE001
                    tails.append(f"--- rank {r}: no output ({e})")
            why = f"did not finish within {timeout} s" if dead_since is None else "lost a rank (its peer was ended after 60 s)"
            raise TimeoutError(f"world-2 exchange ({mode}) {why}\n" + "\n".join(tails))
        time.sleep(0.2)
    for r, p in enumerate(procs):
        outs[r] = p.communicate()[0]
    return [p.returncode for p in procs], outs


@pytest.mark.parametrize("mode", ["signal", "event", "sync", "close_skew"])
def test_rccl_thread_exchange_world2(mode):
    """StepGather(backend="rccl_thread") at WORLD SIZE 2 through agx_exchange_step: 2000 position-task steps per rank
    (1536 envs each, episodes of 37 steps), overlapped (one gather in flight while the next step runs) and synchronous,
    with the device-flag and the event hand-off.  Checked on every rank: the communicator reports 2 ranks; its own slice
    of every gathered buffer is bit-identical to the rows it sent for that step; the checksums of what rank r SENT at
    step t (exchanged over gloo afterwards) equal the checksums of rank r's slice in what EVERY rank RECEIVED for step
    t -- ordering and double buffering over 2000 steps; teardown with the ranks a second apart."""
    steps = 2000 if mode in ("signal", "event") else 400
    codes, outs = _run_world2(mode, steps)
    assert codes == [0, 0], "\n".join(o[-1500:] for o in outs)
    assert all("ok %d steps" % steps in o for o in outs)


@pytest.mark.parametrize("mode", ["signal", "event", "sync", "close_skew"])
def test_peer_push_exchange_world2(mode):
    """StepGather(backend="peer_push") at WORLD SIZE 2, two processes on the one GPU, NO test double: each rank's receive
    buffer and arrival flags are mapped into the other process through hipIpcMemHandle (the path the 8-GPU job takes, there
    over xGMI), the rows are stored there by the observation kernel itself (16-byte quarters of the 64-byte row), the arrival flags are
    raised by the next step's first kernel -- no collective and no extra kernel per step.  Same checks as the RCCL path: 2000 steps, overlapped and synchronous, flag and event hand-off for the
    producer side; every rank's own slice bit-identical to what it sent; per-step checksums of what rank r sent equal those
    of rank r's slice in what every rank received (ordering, the four receive slots, the double-buffered rows)."""
    steps = 2000 if mode in ("signal", "event") else 400
    codes, outs = _run_world2(mode, steps, extra_env={"AGX_TEST_EXCHANGE_BACKEND": "peer_push"})
    assert codes == [0, 0], "\n".join(o[-1500:] for o in outs)
    assert all("ok %d steps" % steps in o for o in outs)


@pytest.mark.parametrize("mode", ["signal", "sync"])
def test_peer_push_through_the_copy_kernel_world2(mode):
    """The other producer of the peer push (what the wide rows of the sensor tasks take, `kernel_push=False`): the observation
    kernel stores its rows locally and raises step_signal, the library's worker thread launches k_push_rows, which copies them
    into both processes' receive buffers (16-byte coalesced) and raises the arrival flags; the consumer orders itself behind
    the local copy with an event and behind the peer's with the flag.  Real IPC mappings, same checks."""
    steps = 1000 if mode == "signal" else 400
    codes, outs = _run_world2(mode, steps, extra_env={"AGX_TEST_EXCHANGE_BACKEND": "peer_push", "AGX_TEST_KERNEL_PUSH": "0"})
    assert codes == [0, 0], "\n".join(o[-1500:] for o in outs)
    assert all("ok %d steps" % steps in o for o in outs)


def test_rccl_thread_exchange_world2_peer_failure_is_an_error_not_a_hang():
    """Rank 1's 50th collective fails (injected): rank 1 raises from the exchange; rank 0's rendezvous ends with an
    error as well (the double marks the segment failed; a silent peer would hit the 5 s bound) -- both processes end."""
    codes, outs = _run_world2("fail", 400, extra_env={"AGX_FAKERCCL_FAIL_RANK": "1", "AGX_FAKERCCL_FAIL_AT": "50",
                                                     "AGX_FAKERCCL_TIMEOUT_S": "5"}, timeout=120)
    assert codes == [3, 3], "\n".join(o[-1500:] for o in outs)
    assert all("exchange failed as expected" in o for o in outs)
