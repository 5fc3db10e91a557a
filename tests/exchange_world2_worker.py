"""Worker of tests/test_gpu_world2_exchange.py::test_{rccl_thread,peer_push}_exchange_world2_*: one of TWO processes sharing the box's
single GPU (AGX_TEST_EXCHANGE_BACKEND=peer_push: the rows travel through peer-mapped memory, no RCCL and no double involved).  torch.distributed runs on gloo (control traffic only); the library-side exchange (csrc/agx_exchange.hip:
worker thread, communication stream, device-flag hand-off, done events) binds tests/fakerccl/libfakerccl.so through
StepGather(rccl_library=...) (path in AGX_TEST_FAKERCCL, read HERE, not by the product), a stream-ordered shared-memory all-gather, because RCCL refuses two ranks on one device.

argv: rank world port mode steps   (mode: signal | event | sync | fail | close_skew)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

rank, world, port, mode, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
import datetime  # noqa: E402
import faulthandler  # noqa: E402

faulthandler.dump_traceback_later(330, exit=False)  # a stuck rank says where (the harness prints this output on its time-out)
dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=240))  # a dead peer is an error, not a 30-minute wait
torch.cuda.set_device(0)
DEV = "cuda:0"

import aerial_gym_simulator_amd  # noqa: E402,F401
from aerial_gym_simulator_amd.config.task_config import position_setpoint_task_config as cfg  # noqa: E402
from aerial_gym_simulator_amd.registry.task_registry import task_registry  # noqa: E402
from aerial_gym_simulator_amd.sharding import StepGather  # noqa: E402

n = 1536  # per rank (an all-gather wants equal counts)
cfg.device, cfg.controller_name, cfg.episode_len_steps = DEV, "lee_position_control", 37
cfg.args = {"shard_rank": rank}
task = task_registry.make_task("position_setpoint_task", seed=10 + rank, num_envs=n, headless=True)
task.reset()
d = task.task_obs["observations"].shape[1]
ready = "event" if mode == "event" else "signal"
BACKEND = os.environ.get("AGX_TEST_EXCHANGE_BACKEND", "rccl_thread")  # "peer_push": real hipIpcMemHandle mapping between the two processes
KERNEL_PUSH = {"0": False, "1": True}.get(os.environ.get("AGX_TEST_KERNEL_PUSH", ""), None)  # None: by row size (here: the kernels push)
sg = StepGather(n, d, DEV, env=task.sim_env, reward=task.rewards, backend=BACKEND, ready=ready, kernel_push=KERNEL_PUSH,
                rccl_library=os.environ.get("AGX_TEST_FAKERCCL") or None)
assert BACKEND != "peer_push" or sg._kernel_push == (KERNEL_PUSH is not False)
assert sg.backend == BACKEND
if BACKEND == "peer_push":  # both processes stored a word and a flag through the other's mapping and saw the other's arrive
    assert sg.push_selftest == "passed", sg.push_selftest
assert sg.comm_info() == (rank, world), sg.comm_info()  # what the communicator itself reports
if ready == "signal":
    assert sg.signal is not None, "the device-flag hand-off was not available (probe failed)"
overlap = mode != "sync"
g = torch.Generator(device=DEV).manual_seed(100 + rank)
actions = [torch.rand(n, 4, device=DEV, generator=g) * 2 - 1 for _ in range(8)]
W = d + 3
own_sum = torch.zeros(steps, dtype=torch.int64, device=DEV)       # checksum of the rows THIS rank sent at step t
got_sum = torch.zeros(steps, world, dtype=torch.int64, device=DEV)  # checksum of every rank's slice received for step t
own_ok = torch.ones((), dtype=torch.bool, device=DEV)
lag = sg.lag if overlap else 0  # the overlapped form hands back the rows of `lag` steps ago (2 when the kernels push the rows themselves)
history = []
try:
    for t in range(steps):
        obs, rew, term, trunc, _ = task.step(actions[t % 8])
        p = task.sim_env._parity
        out = sg.exchange(p, overlap=overlap)
        # stream-ordered bookkeeping, no host synchronisation inside the loop
        mine = torch.cat([obs["observations"], rew[:, None], term[:, None].float(), trunc[:, None].float()], dim=1)
        own_sum[t] = mine.view(torch.int32).long().sum()  # bit patterns, exact and order-independent
        ref_t = t - lag
        history.append(mine)
        if out is not None:
            assert ref_t >= 0
            got_sum[ref_t] = out.view(torch.int32).view(world, n, W).long().sum(dim=(1, 2))
            own_ok &= torch.equal(out.view(world, n, W)[rank], history[ref_t - t - 1])
        elif overlap:
            assert t < lag
        history = history[-3:]
        if mode == "fail" and t % 16 == 0:
            torch.cuda.synchronize()
    sg.flush()
    torch.cuda.synchronize()
except RuntimeError as e:
    if mode == "fail":
        print(f"rank {rank}: exchange failed as expected: {str(e)[:160]}", flush=True)
        os._exit(3)  # the peer may be gone: no collective teardown
    raise
if mode == "fail":
    print(f"rank {rank}: no failure surfaced", flush=True)
    os._exit(4)
assert bool(own_ok), "own slice of the gathered buffer differs from the rows this rank sent"
last = steps - lag
# every rank's checksum of what it SENT, exchanged over gloo, against what each rank RECEIVED
sent = [torch.zeros(steps, dtype=torch.int64) for _ in range(world)]
dist.all_gather(sent, own_sum.cpu())
got = got_sum.cpu()
for r in range(world):
    assert torch.equal(got[:last, r], sent[r][:last]), (rank, r, "received rows are not the rows rank %d sent, step by step" % r)
assert int(task.sim_env.global_tensor_dict["episode_count"].sum()) > n  # resets happened along the way
if mode == "close_skew" and rank == 1:
    time.sleep(1.0)  # the ranks tear their communicators down a second apart
sg.close()
dist.barrier()
dist.destroy_process_group()
print(f"rank {rank}: ok {steps} steps mode {mode}", flush=True)
