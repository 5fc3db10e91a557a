#!/usr/bin/env python
"""Headline benchmark: env-steps/sec of task.step() on synthetic actions.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload dynamics|depth]

N = 1 : BASELINE.json configs[1] -- base_quadrotor, position-setpoint task, Lee position
        controller, 8192 envs, empty_env, dynamics only -- plus (extra keys, same JSON line)
        the "+depth" number of configs[2] when --with-depth is given.
N > 1 : one process per GPU (torch.distributed / RCCL), 8192 envs per rank (weak scaling),
        one all_gather of the packed (obs | reward | termination | truncation) per step.

A "step" is one task.step(): k physics sub-steps + reward + reset + observation for every
env.  Inputs (state, scenes, actions) are resident in HBM before the timed region.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# algorithmic bytes per env per env-step, dynamics-only quad (A=4, M=4, obs 13, k=1)
#   4*(13 state in + 4 action + 4 thrust in + 13 state out + 4 thrust out) = 152 B   -> agx_dynamics_substeps
#   4*(13 obs + 1 reward) + 2 flags                                       =  58 B   -> reward / obs kernels
#   fused: k_env_step moves 152 + 4 (reward) + 2 (flags) = 158 B, k_reset_masked<.., obs> the 52 B observation
BYTES_DYNAMICS_KERNEL = 158
BYTES_ENV_STEP = 210
# the reset / observation launch on its own (kernel view): reads 13 state + 6 body-frame velocity + 3 target floats + 3 flag bytes,
# writes the 13-float observation: 4 * (13 + 6 + 3 + 13) + 3 = 143 B per env (52 of them are the step's algorithmic observation bytes)
BYTES_RESET_OBS_KERNEL = 143


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--regions", type=int, default=5, help="timed regions of --steps steps each; `value` is their median")
    ap.add_argument("--num-envs", type=int, default=None, help="envs per GPU (default: 8192; 4096 for the LiDAR workloads, BASELINE configs[3])")
    ap.add_argument("--workload", default="dynamics", choices=["dynamics", "depth", "lidar", "lidar_velocity", "lidar_nav"])
    ap.add_argument("--no-depth", action="store_true", help="skip the +depth config (BASELINE configs[2]) extra keys")
    ap.add_argument("--no-lidar", action="store_true", help="skip the LiDAR config (BASELINE configs[3]) extra keys")
    ap.add_argument("--no-strict", action="store_true", help="skip the strict_rng (reference torch RNG stream) leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sync-gather", action="store_true", help="N > 1: wait for each step's all-gather before the next step")
    ap.add_argument("--exchange", default="auto", choices=["auto", "process_group", "rccl_thread", "peer_push"],
                    help="N > 1: who enqueues the per-step all-gather. auto = time both (torch's process group first, then the "
                         "library's RCCL worker thread under a watchdog) and report the faster one as `value`")
    ap.add_argument("--strict-rng", action="store_true", help="reference-faithful RNG consumption (host sync per step)")
    ap.add_argument("--exchange-selftest-only", action="store_true",
                    help="no timing: build every exchange backend (process_group, peer_push, rccl_thread) on the N ranks, run "
                         "--selftest-steps checksum-verified exchange steps on each and print ONE JSON verdict line per backend")
    ap.add_argument("--selftest-steps", type=int, default=1000)
    ap.add_argument("--preflight-steps", type=int, default=200,
                    help="N > 1: checksum-verified exchange steps per backend BEFORE anything is timed (0 = skip); the verdicts go to "
                         "stderr and into the line's exchange.preflight")
    args = ap.parse_args()
    if args.num_envs is None:
        args.num_envs = 4096 if args.workload in ("lidar", "lidar_velocity") else 8192
    return args


LEAN_AT_SCALE_ENVS = (1 << 21) + 256  # its own grid size: the PMC file keys kernels by name + grid, and the lean launch is the same kernel


def make_task(workload, num_envs, device, strict_rng, rank=0, obstacles="all", lean=None, extra_args=None):
    """obstacles: "all" = every obstacle of the scene is in the env (BASELINE configs 3/4: 100 boxes + 6 walls);
    "curriculum" = the task's own curriculum start (navigation_task_config.py: level 15 of 106).
    extra_args: merged into the task's `args` dict (profiles/ scripts: {"step_graph": True}, {"bvh_box_objects": False}, ...)."""
    extra = dict(extra_args or {})
    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.config.task_config import navigation_task_config, position_setpoint_task_config
    from aerial_gym_simulator_amd.registry.task_registry import task_registry

    if workload == "dynamics":
        cfg = position_setpoint_task_config
        cfg.controller_name = "lee_position_control"
        cfg.device = device
        cfg.args = dict({"strict_rng": strict_rng, "shard_rank": rank}, **extra)
        if lean is not None:  # the lean step is opt-in (args={"lean_step": True}); None = the task's default = every tensor maintained
            cfg.args["lean_step"] = bool(lean)
        return task_registry.make_task("position_setpoint_task", seed=1 + rank, num_envs=num_envs, headless=True)
    if workload == "lidar_nav":  # the reference's LiDAR-navigation recipe (magpie, 48 x 120 dome LiDAR, 337-D obs)
        from aerial_gym_simulator_amd.config.task_config import lidar_navigation_task_config as lcfg

        lcfg.device = device
        lcfg.args = dict({"strict_rng": strict_rng, "shard_rank": rank}, **extra)
        return task_registry.make_task("lidar_navigation_task", seed=1 + rank, num_envs=num_envs, headless=True)
    if workload == "lidar":  # BASELINE configs[3] as written: FULLY-ACTUATED octarotor (7-D command) + 32 x 512 LiDAR
        from aerial_gym_simulator_amd.config.task_config import fully_actuated_lidar_navigation_task_config as cfg
    else:
        cfg = navigation_task_config
    if not hasattr(cfg, "_reference_curriculum"):
        cfg._reference_curriculum = (cfg.curriculum.min_level, cfg.curriculum.max_level)
    cfg.curriculum.min_level, cfg.curriculum.max_level = (106, 107) if obstacles == "all" else cfg._reference_curriculum
    cfg.device = device
    cfg.args = dict({"strict_rng": strict_rng, "shard_rank": rank}, **extra)  # rank: own scenes, RNG stream and semantic-id range
    if workload == "lidar":
        return task_registry.make_task("navigation_task_fully_actuated_lidar", seed=1 + rank, num_envs=num_envs, headless=True)
    if workload == "lidar_velocity":  # the same robot and sensor under the Lee velocity controller (4-D command)
        cfg.robot_name = "base_octarotor_with_lidar_32x512"
        cfg.controller_name = "octarotor_velocity_control"
    else:
        cfg.robot_name, cfg.controller_name = "base_quadrotor_with_camera_64x48", "lee_velocity_control"
    return task_registry.make_task("navigation_task", seed=1 + rank, num_envs=num_envs, headless=True)


def timed_steps(task, actions, steps, warmup, world, gather_buf=None, overlap=True):
    import torch.distributed as dist

    env = task.sim_env

    def one(i):
        task.step(actions[i % len(actions)])
        if gather_buf is not None:
            # ONE RCCL all-gather per env step of the rows the observation kernel wrote (obs | reward |
            # terminated | truncated of every env of every rank); with `overlap` it stays in flight
            # while the next step computes and the previous step's result is handed out
            gather_buf.exchange(env._parity, overlap=overlap)

    # set-up has finished on the device before the warm-up starts: the launches that follow a process's FIRST device
    # synchronize are slow (+0.09 ms over the next ~20 steps, profiles/r02_short_run_probe.txt); without this line that first
    # synchronize is the one that opens the timed region, and a 20-step run reads 16.8 instead of 12.7 us per step
    torch.cuda.synchronize()
    for i in range(warmup):
        one(i)
    if gather_buf is not None:
        gather_buf.flush()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        one(i)
    timed_steps.last_host_s = time.perf_counter() - t0  # the host's share: all launches of the region enqueued (rank-local clock)
    if gather_buf is not None:
        gather_buf.flush()  # the last collective is inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def desynchronise_episodes(task, seed=4321):
    """Episodes start together at reset(): all envs would truncate in the same step and, until then, in none -- a short timed
    region right after reset() never sees a reset.  The steady state of a long-running job is episodes
    spread uniformly over their length: sim_steps <- U{0 .. episode_len - 1}, written through the public tensor the reference
    exposes too (env_manager.py: `self.sim_steps`); from the next step on, num_envs / episode_len envs truncate in EVERY step."""
    env = task.sim_env
    L = int(task.task_config.episode_len_steps)
    steps = env.global_tensor_dict["sim_steps"]
    g = torch.Generator(device=steps.device).manual_seed(seed + int(getattr(env, "env_offset", 0)))
    steps[:] = torch.randint(0, L, (env.num_envs,), device=steps.device, generator=g, dtype=torch.int32)
    torch.cuda.synchronize()
    return {"episode_len_steps": L, "expected_truncations_per_step": env.num_envs / L,
            "how": "sim_steps <- U{0..episode_len-1} after reset(), before the warm-up"}


def resets_per_step(task, before, steps):
    """episodes that ended inside a region, per step (episode_count is bumped by the reset kernels): what the timed steps contained"""
    now = int(task.sim_env.global_tensor_dict["episode_count"].sum().item())
    return (now - before) / max(steps, 1), now


def hbm_copy_gbs(device, nbytes=1 << 30, reps=10):
    """Measured HBM bandwidth of a float4 streaming copy (read + write bytes / time; the library's own HIP kernel,
    agx_copy_f4) on this box: the achievable ceiling next to the 8 TB/s of specification (MI355X_MICROARCH.md: 6.29 TB/s)."""
    import ctypes as C

    from aerial_gym_simulator_amd import _lib

    lib = _lib.load()
    a = torch.empty(nbytes // 4, device=device)
    b = torch.empty_like(a)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 0.0
    for _ in range(3):
        _lib.check(lib.agx_copy_f4(_lib.dptr(a), _lib.dptr(b), nbytes, stream), "agx_copy_f4")
        start.record()
        for _ in range(reps):
            lib.agx_copy_f4(_lib.dptr(a), _lib.dptr(b), nbytes, stream)
        stop.record()
        torch.cuda.synchronize()
        best = max(best, 2.0 * nbytes * reps / (start.elapsed_time(stop) * 1e-3) / 1e9)
    return best


_PMC = None


def _pmc():
    """The committed rocprofv3 PMC passes (profiles/pmc_traffic.json, written by profiles/collect_pmc.py): HBM bytes and
    vector instructions per launch.  Counters cannot be collected inside a timed run, so they are read from the file --
    together with the hash of the kernel sources they were measured on: when the sources changed since, every number taken
    from the file is reported with "stale": true instead of silently."""
    global _PMC
    if _PMC is None:
        path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        try:
            _PMC = json.load(open(path))
        except Exception:  # noqa: BLE001
            _PMC = {}
        from aerial_gym_simulator_amd import _lib

        # the counters were collected on a library that reported this build id; `stale` is about the BINARY now loaded
        # (agx_build_id: hash of the sources and flags it was compiled from), not about the files lying next to it
        _PMC["_stale"] = _PMC.get("build_id", _PMC.get("source_hash")) != _lib.build_id()
    return _PMC


def pmc_traffic(tag):
    return _pmc().get(tag)


def pmc_stale():
    return bool(_pmc().get("_stale", True))


VALU_PEAK_GUIDE = 256 * 4 * 2.4e9 / 2.0  # MI355X_MICROARCH.md: v_fma_f32 wave64 = 2 cycles on a SIMD-32, 1024 SIMDs, 2.4 GHz: 1.229 T/s


def valu_peak():
    """Plain (non-packed) fp32 VALU issue ceiling in wave64 instructions/s, two readings:
    * measured on an MI355X by profiles/src/valu_peak.hip (profiles/r04_valu_peak.json: 16 independent chains per wave, scalar /
      inline-constant second and third operands, 8 waves per SIMD) -- what the chip sustains under its power budget (the
      effective clock of that run, GRBM_GUI_ACTIVE / wall, is recorded next to it);
    * the hardware guide's 2 cycles per wave64 instruction per SIMD at the 2.4 GHz maximum clock: 1.229 T/s."""
    for name in ("r04_valu_peak.json", "r03_valu_peak.json", "r02_valu_peak.json"):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
            return float(d["wave_instr_per_s"]), f"measured (profiles/{name})", d.get("effective_clock_ghz")
        except Exception:  # noqa: BLE001
            continue
    return VALU_PEAK_GUIDE, "guide: 2 cycles per wave64 instruction per SIMD at 2.4 GHz", None


def valu_roofline(kernel_key, launch_s):
    """Vector instructions the kernel issues per launch (SQ_INSTS_VALU, committed PMC pass) over the launch duration
    measured live, against both VALU issue ceilings."""
    n_valu = _pmc().get("valu_wave_instructions", {}).get(kernel_key)
    peak, how, clk = valu_peak()
    out = {"unit": "G wave64-instr/s", "peak_measured": peak / 1e9, "peak_measured_source": how, "peak_measured_effective_clock_ghz": clk,
           "peak_guide": VALU_PEAK_GUIDE / 1e9, "peak_guide_source": "MI355X_MICROARCH.md: 2 cycles per wave64 v_fma_f32 per SIMD-32, 1024 SIMDs, 2.4 GHz"}
    if n_valu is None:
        return dict(out, achieved=None, frac_of_measured=None, frac_of_guide=None)
    ach = n_valu / launch_s
    out = dict(out, achieved=ach / 1e9, frac_of_measured=ach / peak, frac_of_guide=ach / VALU_PEAK_GUIDE,
               valu_wave_instructions_per_launch=n_valu, stale=pmc_stale())
    busy = valu_busy_time(kernel_key, n_valu, peak)
    if busy:
        out["pipe_busy"] = dict(busy, frac_of_launch=busy["seconds_lower_bound"] / launch_s)
    return out


def valu_busy_time(kernel_key, n_valu, plain_rate):
    """How long the vector pipes are busy per launch, from the kernel's instruction CLASSES (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32 /
    _F64, _INT32, _INT64, _CVT: committed PMC passes) times the issue rate of each class as measured on this GPU
    (profiles/src/valu_peak.hip: v_fma_f64, v_mul / v_add_f64, v_rcp / v_sqrt_f32, v_rcp_f64, the f32 <-> f64 conversions; same
    16-chain, 8-waves-per-SIMD shape as the plain-fp32 ceiling).  An instruction COUNT against the plain-fp32 ceiling says how
    many slots were used; this says how much of the launch the pipe had no slot left -- for the env-step kernels, whose
    elementary functions are evaluated in float64, the two differ.  A lower bound: integer multiplies (Philox) are counted at the
    plain rate with the rest of INT32, and whatever no class counter sees (moves, selects, compares, DPP) likewise."""
    mix = (_pmc().get("sq_breakdown", {}).get(kernel_key) or {}).get("valu_class_mix")
    if not mix:
        return None
    rates = None
    for name in ("r04_valu_peak.json",):
        try:
            rates = json.load(open(os.path.join(ROOT, "profiles", name))).get("class_wave_instr_per_s")
        except Exception:  # noqa: BLE001
            rates = None
    if not rates:
        return None
    def rate(cls):
        r = rates.get(cls)
        return (sum(r.values()) / len(r)) if r else plain_rate
    cls_rate = {"ADD_F64": rate("MUL_F64+ADD_F64"), "MUL_F64": rate("MUL_F64+ADD_F64"), "FMA_F64": rate("FMA_F64"),
                "TRANS_F32": rate("TRANS_F32"), "TRANS_F64": rate("TRANS_F64"), "CVT": rate("CVT")}
    t, by = 0.0, {}
    for cls, count in mix.items():
        r = cls_rate.get(cls, plain_rate)
        by[cls.split(" ")[0]] = {"wave_instr": count, "rate_g_per_s": r / 1e9, "us": count / r * 1e6}
        t += count / r
    return {"seconds_lower_bound": t, "us_lower_bound": t * 1e6, "by_class": by,
            "plain_only_us": n_valu / plain_rate * 1e6}


def roofline_block(kernel, launch_s, algorithmic_bytes, key, copy_gbs=None, timing=None, note=None, bound="hbm", **extra):
    """The contract's `roofline` object for one kernel: bound "hbm" -- achieved = algorithmic bytes per launch / average launch
    duration measured live with HIP events on the launch stream, peak = 8 TB/s (specification), traffic = HBM bytes per launch
    from the committed PMC passes -- with the achievable copy rate of this box and the kernel's vector-instruction issue rate
    against both VALU ceilings next to it (`valu`): none of these kernels is HBM bound, and that object says what bounds them."""
    ach = algorithmic_bytes / launch_s / 1e9
    tr = pmc_traffic(key)
    blk = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": tr,
           "traffic_stale": pmc_stale(), "kernel": kernel, "launch_us": launch_s * 1e6,
           "algorithmic_bytes_per_launch": algorithmic_bytes}
    if tr:
        blk["traffic_over_algorithmic"] = tr / algorithmic_bytes
        blk["traffic_rate_gbs"] = tr / launch_s / 1e9
    if copy_gbs:
        blk["peak_measured_copy"] = copy_gbs
        blk["frac_of_measured_copy"] = ach / copy_gbs
        if tr:
            blk["traffic_rate_over_measured_copy"] = blk["traffic_rate_gbs"] / copy_gbs
    if timing:
        blk["launch_us_detail"] = {k: (v * 1e6 if isinstance(v, float) else v) for k, v in timing.items()}
    blk["valu"] = valu_roofline(key, launch_s)
    if note:
        blk["note"] = note
    blk.update(extra)
    if bound == "valu" and blk["valu"].get("achieved") is not None:
        # the bound that BINDS this kernel is vector-instruction issue, not HBM: the object's headline numbers are the VALU ones
        # (instructions per launch from the committed PMC pass over the live launch time, against the guide's issue ceiling);
        # the byte view stays next to it under "hbm"
        v = blk["valu"]
        hbm = {k: blk[k] for k in ("achieved", "peak", "unit", "frac", "traffic", "traffic_stale", "algorithmic_bytes_per_launch",
                                   "traffic_over_algorithmic", "traffic_rate_gbs", "peak_measured_copy", "frac_of_measured_copy",
                                   "traffic_rate_over_measured_copy") if k in blk}
        blk.update(bound="valu", achieved=v["achieved"], peak=v["peak_guide"], unit=v["unit"], frac=v["frac_of_guide"],
                   frac_of_measured_peak=v["frac_of_measured"], counters_stale=v.get("stale"), hbm=hbm)
    elif bound == "valu":
        blk["bound_note"] = "vector-issue bound by design; no instruction count for this kernel instance in profiles/pmc_traffic.json, the byte view is shown"
    return blk


def env_step_key(task, k):
    """key of the env-step kernel instance in the PMC file (template arguments + grid size in threads), as the library
    names the instance it launches for these arguments"""
    import ctypes as C

    env = task.sim_env
    buf = C.create_string_buffer(128)
    env._lib.agx_env_step_kernel(env._params, env._buffers, env.num_envs, int(k), env.task_args, buf, 128)
    return buf.value.decode()


def raycast_key(task):
    """key of the frame's ray-cast kernel instance in the PMC file (template arguments <lidar, variant> + grid size in threads),
    as the library names the instance it launches for these sizes"""
    import ctypes as C

    from aerial_gym_simulator_amd import _lib

    sen = task.sim_env.robot_manager.warp_sensor
    cfg = sen.cfg
    variant = 2 if sen.is_stereo else (1 if sen.is_normal else 0)
    buf = C.create_string_buffer(128)
    _lib.check(_lib.load().agx_raycast_kernel(task.num_envs, cfg.num_sensors, cfg.width, cfg.height, int(sen.is_lidar), variant, buf, 128), "agx_raycast_kernel")
    return buf.value.decode()


def cpu_baseline_reference():
    """The reference's own torch CPU path (oracle/time_reference_cpu.py).  Where the reference tree is present on THIS box
    (AERIAL_GYM_REFERENCE_ROOT, default /root/reference) it is timed here and now, on this box's host cores -- the comparison
    north_star names; otherwise the committed measurement of the build container is echoed (its `host` field says so)."""
    root = os.environ.get("AERIAL_GYM_REFERENCE_ROOT") or os.environ.get("AERIAL_GYM_REFERENCE") or "/root/reference"
    os.environ["AERIAL_GYM_REFERENCE_ROOT"] = root  # (what oracle/ref_shells.py reads, in the child process below)
    keys = ("value", "unit", "cores", "kind", "sample", "host", "cpu_model", "what")
    if os.path.isdir(os.path.join(root, "aerial_gym")):
        import subprocess
        import tempfile

        try:
            with tempfile.TemporaryDirectory() as tmp:
                dst = os.path.join(tmp, "ref.json")
                subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "time_reference_cpu.py"), "--out", dst, "--seconds", "12",
                                "--host", "this box (the bench host's cores)"], check=True, capture_output=True, timeout=300)
                d = json.load(open(dst))
            return dict({k: d[k] for k in keys}, measured_here=True)
        except Exception as e:  # noqa: BLE001
            print(f"[bench] the reference tree is present but could not be timed ({type(e).__name__}: {e})", file=sys.stderr)
    for name in ("r06_cpu_baseline_reference.json", "r05_cpu_baseline_reference.json", "r04_cpu_baseline_reference.json", "r03_cpu_baseline_reference.json"):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
            return dict({k: d[k] for k in keys}, measured_here=False, source="profiles/" + name)
        except Exception:  # noqa: BLE001
            continue
    return None


def live_parity(device):
    """One golden case through the C ABI right here, next to the timing: base_quadrotor + lee_position_control, the
    reference's recorded inputs -> its recorded outputs (tests/golden/step_quad_position.npz, 6 sub-steps x 64 envs).
    No CPU code is involved: the comparison is GPU vs the reference's own numbers."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    from conftest import elem_err, golden_params, load_golden, max_abs
    from gpu_harness import DynHarness

    g = load_golden("step_quad_position")
    pd = golden_params(g)
    n, K = g["state"].shape[1], g["state"].shape[0]
    H = DynHarness(pd, n, dev=device)
    H.set(kT=g["kT"], tau_inc=g["tau_inc"], tau_dec=g["tau_dec"])
    H.set_gains(g["Kp"], g["Kv"], g["KR"], g["Kw"])
    out = {"state_next": 0.0, "wrench": 0.0, "thrust_over_full_scale": 0.0, "body_rates": 0.0}
    # the same case made by the reference's code with correctly rounded elementary functions (tests/golden/cr/): bit for bit
    gc = load_golden("step_quad_position", cr=True)
    H.set(kT=gc["kT"], tau_inc=gc["tau_inc"], tau_dec=gc["tau_dec"])
    H.set_gains(gc["Kp"], gc["Kv"], gc["KR"], gc["Kw"])
    exact = True
    for k in range(K):
        H.set(state=gc["state"][k], thrust=gc["thrust_in"][k])
        H.substeps(gc["action"][k], 1)
        exact = exact and np.array_equal(H.get("wrench"), gc["wrench_cmd"][k]) and np.array_equal(H.get("thrust"), gc["thrust_out"][k])
        if k + 1 < K:
            exact = exact and np.array_equal(H.get("state"), gc["state"][k + 1])
    H.set(kT=g["kT"], tau_inc=g["tau_inc"], tau_dec=g["tau_dec"])
    H.set_gains(g["Kp"], g["Kv"], g["KR"], g["Kw"])
    for k in range(K):
        H.set(state=g["state"][k], thrust=g["thrust_in"][k])
        H.substeps(g["action"][k], 1)
        if k + 1 < K:
            out["state_next"] = max(out["state_next"], elem_err(H.get("state"), g["state"][k + 1]))
        out["wrench"] = max(out["wrench"], elem_err(H.get("wrench"), g["wrench_cmd"][k]))
        out["thrust_over_full_scale"] = max(out["thrust_over_full_scale"], max_abs(H.get("thrust"), g["thrust_out"][k]) / pd["max_thrust"])
        out["body_rates"] = max(out["body_rates"], elem_err(H.get("derived")[:, 10:16], np.concatenate([g["vbody"][k], g["wbody"][k]], axis=1)))
    return {"case": "tests/golden/step_quad_position.npz (reference BaseMultirotor.step outputs, 6 sub-steps x 64 envs)",
            "max_err_vs_reference": out, "bit_exact_vs_reference_with_correctly_rounded_functions": bool(exact), "unit": "|err| / max(1, |x|) per component (thrust: / 2 N full scale)",
            "gates": "tests/ (pytest -m gpu): bit-exact vs the CPU oracle, <= 1e-5 vs the reference (every state component), bit-exact vs the reference with correctly rounded elementary functions; "
                     "measured maxima of the last full run: profiles/r06_parity_report.json"}


def kernel_time_dynamics(task, actions, reps=400):
    """Duration of the env-step kernel, HIP events on the stream it is launched on, three ways (all reported; rocprofv3's
    per-kernel average over the same command lies between the first two, profiles/r04_*_kernel_stats.csv):

    * `in_step`  -- the launch bracketed by two events INSIDE real task steps (env-step launch, then the reset / observation
      launch, exactly the sequence of the timed region), the queue pre-filled behind a blocker so that the host is out of
      the picture: from the completion of the preceding kernel to the completion of this one.  This is the figure the
      roofline is priced with: it is what the kernel costs the step.
    * `back_to_back` -- the same launch repeated `reps` times in a row, total / reps (round 2's figure: it depends on how
      well consecutive dispatches of the SAME kernel overlap, and did not reproduce between runs).
    * `isolated` -- one launch on an idle device between two events (cold L2, dispatch latency in full).
    `event_pair` is what two events with nothing in between measure (their own cost)."""
    from aerial_gym_simulator_amd import _lib

    env = task.sim_env
    a = actions[0].contiguous()
    k = env.num_physics_steps()
    lib, P, B, n, ptr = env._lib, env._params, env._buffers, env.num_envs, _lib.dptr(a)
    blocker = torch.randn(4096, 4096, device=a.device)
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = {}
    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        for _ in range(6):
            blocker @ blocker  # keeps the GPU busy while the host enqueues the launches below
        start.record()
        s = env._stream()
        for _ in range(reps):
            lib.agx_env_step(P, B, n, ptr, k, env.task_args, s)
        stop.record()
        torch.cuda.synchronize()
        ms = start.elapsed_time(stop) / reps
        best = ms if best is None else min(best, ms)
    out["back_to_back"] = best * 1e-3
    # isolated launches and the cost of an event pair
    iso, pair = [], []
    for _ in range(40):
        torch.cuda.synchronize()
        start.record()
        lib.agx_env_step(P, B, n, ptr, k, env.task_args, env._stream())
        stop.record()
        torch.cuda.synchronize()
        iso.append(start.elapsed_time(stop))
        start.record()
        stop.record()
        torch.cuda.synchronize()
        pair.append(start.elapsed_time(stop))
    iso.sort(), pair.sort()
    out["isolated"] = iso[len(iso) // 2] * 1e-3
    out["event_pair"] = pair[len(pair) // 2] * 1e-3
    # inside real steps: the general (two host calls) path of task.step() issues the same two launches as the fast path;
    # its env-step call is wrapped so that every launch sits between two events
    m = min(reps, 300)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(m)]
    evs2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(m)]
    real_env_step, real_post = lib.agx_env_step, lib.agx_post_step_position
    idx, idx2 = [0], [0]

    class _Timed:
        def __getattr__(self, name):
            return getattr(lib, name)

        def agx_env_step(self, *args):
            e0, e1 = evs[idx[0]]
            e0.record()
            rc = real_env_step(*args)
            e1.record()
            idx[0] += 1
            return rc

        def agx_post_step_position(self, *args):  # the step's second launch: reset of the flagged envs + observation (+ exchange rows)
            e0, e1 = evs2[idx2[0]]
            e0.record()
            rc = real_post(*args)
            e1.record()
            idx2[0] += 1
            return rc

    plan, task._plan = getattr(task, "_plan", None), None  # general path: env.step + reward + post_reward_calculation_step
    env._lib = _Timed()
    try:
        torch.cuda.synchronize()
        for _ in range(6):
            blocker @ blocker
        for i in range(m):
            task.step(actions[i % len(actions)])
        torch.cuda.synchronize()
    finally:
        env._lib = lib
        task._plan = plan
    d = sorted(e0.elapsed_time(e1) for e0, e1 in evs[: idx[0]])
    if d:
        out["in_step"] = sum(d) / len(d) * 1e-3
        out["in_step_min"], out["in_step_median"], out["in_step_max"] = d[0] * 1e-3, d[len(d) // 2] * 1e-3, d[-1] * 1e-3
        out["in_step_samples"] = len(d)
    out["primary"] = out.get("in_step", out["back_to_back"])
    d2 = sorted(e0.elapsed_time(e1) for e0, e1 in evs2[: idx2[0]])
    post = None
    if d2:
        post = {"in_step": sum(d2) / len(d2) * 1e-3, "in_step_min": d2[0] * 1e-3, "in_step_median": d2[len(d2) // 2] * 1e-3,
                "in_step_max": d2[-1] * 1e-3, "in_step_samples": len(d2), "event_pair": out["event_pair"]}
        post["primary"] = post["in_step"]
    out["_post_step"] = post
    return out, k


def kernel_time_raycast(task, reps=20):
    """Average duration of one ray-cast launch (all envs, one frame), HIP events on the launch stream."""
    env = task.sim_env
    sensor = env.robot_manager.warp_sensor
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        start.record()
        s = env._stream()
        for _ in range(reps):
            sensor.raycast(s, fuse_limits=sensor.limits_fusable())  # the launch the step issues (range limits in its epilogue)
        stop.record()
        torch.cuda.synchronize()
        ms = start.elapsed_time(stop) / reps
        best = ms if best is None else min(best, ms)
    return best * 1e-3


def raycast_bytes_per_env(task):
    """scene read once (12 V + 12 T + 4 V + 32 (2T - 1)) + pose 28 S + image 4 H W S (+ 4 H W S seg)."""
    sc = task.sim_env.scene
    cfg = task.sim_env.robot_manager.warp_sensor.cfg
    K, T = sc.num_assets, sc.num_tris
    V = 8 * K
    scene = 12 * V + 12 * T + 4 * V + 32 * (2 * T - 1)
    S, H, W = cfg.num_sensors, cfg.height, cfg.width
    img = 4 * H * W * S * (2 if cfg.segmentation_camera else 1) * (3 if cfg.return_pointcloud else 1)
    return scene + 28 * S + img


def exchange_diagnostics(task, actions, args, world, gather_buf, dt_with):
    """N > 1 only (every rank runs it: it contains collectives). Splits the step time into the simulator
    and the per-step all-gather: the same steps without the exchange, and the collective on its own
    (back-to-back = enqueue/throughput cost, synchronised = latency)."""
    import torch.distributed as dist

    steps = max(args.steps // 2, 50)
    dt_without = timed_steps(task, actions, steps, max(args.warmup // 4, 10), world, None)
    src, dst = gather_buf.rows[0], gather_buf.gathered[0]
    for _ in range(20):
        dist.all_gather_into_tensor(dst, src)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps):
        dist.all_gather_into_tensor(dst, src)
    t_enqueue = (time.perf_counter() - t0) / reps
    torch.cuda.synchronize()
    t_back_to_back = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(50):
        dist.all_gather_into_tensor(dst, src)
        torch.cuda.synchronize()
    t_latency = (time.perf_counter() - t0) / 50
    return {"ms_per_step_with_exchange": 1e3 * dt_with / args.steps, "ms_per_step_without_exchange": 1e3 * dt_without / steps,
            "all_gather_us_host_enqueue": 1e6 * t_enqueue, "all_gather_us_back_to_back": 1e6 * t_back_to_back,
            "all_gather_us_synchronised": 1e6 * t_latency, "bytes_per_rank": src.numel() * 4,
            "note": "rank 0's clock; the exchange of step t runs on RCCL's stream while step t+1 computes"}


def library_exchange_leg(args, world, rank, device, out, backend, limit_s=240.0):
    """Another pass of the N > 1 job with the per-step exchange run by the simulator library (csrc/agx_exchange.hip) instead
    of torch's process group: `peer_push` (rows stored straight into the peers' receive buffers, no collective kernel per
    step) or `rccl_thread` (an RCCL all-gather enqueued by the library's worker thread on its own communicator): same tasks,
    same K and W.  The fastest leg becomes `value`; all are reported.  A watchdog prints the line measured so far and ends
    the process if the leg does not finish (a half-built exchange cannot be recovered in-process)."""
    import threading

    import torch.distributed as dist
    from aerial_gym_simulator_amd.sharding import StepGather

    n_gpus = max(world, 1)
    report = out.setdefault("exchange", {})
    report.setdefault("process_group", {"value": out["value"], "ms_per_step": out["ms_per_step"], "ranks_seen": report.get("ranks_seen"),
                                        "plus_depth_value": out.get("plus_depth", {}).get("value"),
                                        "plus_depth_ms_per_step": out.get("plus_depth", {}).get("ms_per_step")})

    res = {}

    def give_up():
        report[backend] = dict(res, error=res.get("error", "") + f" [no agreement of the ranks within {limit_s:.0f} s]")
        if rank == 0:
            emit_line(out, _JSON_FD[0])
        os._exit(0)

    dog = threading.Timer(limit_s, give_up)
    dog.daemon = True
    dog.start()
    stage = "build"
    try:
        torch.cuda.empty_cache()
        task = make_task("dynamics", args.num_envs, device, args.strict_rng, rank)
        task.reset()
        desynchronise_episodes(task)  # the same steady state as the first leg's `value`: resets in every timed step
        N, A = task.num_envs, task.task_config.action_space_dim
        g = torch.Generator(device=device).manual_seed(1234 + rank)
        actions = [torch.rand(N, A, device=device, generator=g) * 2 - 1 for _ in range(16)]
        stage = "setup"  # a backend that cannot be SET UP fails on every rank together (StepGather._agree) and leaves nothing behind
        gb = StepGather(N, task.task_obs["observations"].shape[1], device, env=task.sim_env, reward=task.rewards, backend=backend)
        stage = "run"
        res["backend"] = gb.backend
        res["producer_hand_off"] = "device flag (step_signal)" if gb.signal is not None else "event"
        if backend == "peer_push":
            res["flags_in_uncached_memory"] = gb.push_flags_uncached
            res["connection_selftest"] = gb.push_selftest  # a word + a flag stored through every mapping and checked on arrival
        res["communicator"] = dict(zip(("rank", "ranks"), gb.comm_info()))  # ncclCommUserRank / ncclCommCount, or the ranks whose buffers were mapped
        res["ranks_seen"] = res["communicator"]["ranks"]
        print(f"[bench rank {rank}/{world}] exchange leg {backend}: set up as {gb.backend}, communicator reports rank "
              f"{res['communicator']['rank']} of {res['ranks_seen']}"
              + (f", peer-mapping self-test {gb.push_selftest}, kernel push {gb._kernel_push}" if backend == "peer_push" else ""),
              file=sys.stderr, flush=True)
        if res["communicator"]["ranks"] != max(world, 1):
            raise RuntimeError(f"the exchange communicator spans {res['communicator']['ranks']} ranks, the job has {world}")
        if os.environ.get("AGX_BENCH_INJECT_EXCHANGE_FAILURE") == str(rank):  # exercises the abandon path below
            raise RuntimeError("injected failure")
        dt = timed_steps(task, actions, args.steps, args.warmup, world, gb, overlap=not args.sync_gather)
        res.update(value=n_gpus * N * args.steps / dt, ms_per_step=1e3 * dt / args.steps)
        gb.close()
        del task, gb
        if not args.no_depth:
            torch.cuda.empty_cache()
            t2 = make_task("depth", args.num_envs, device, args.strict_rng, rank)
            t2.reset()
            desynchronise_episodes(t2)
            a2 = [torch.rand(N, 4, device=device, generator=g) * 2 - 1 for _ in range(4)]
            s2 = min(max(args.steps // 10, 20), 300)
            gb2 = StepGather(N, t2.task_obs["observations"].shape[1], device, env=t2.sim_env, reward=t2.rewards, backend=backend)
            dt2 = timed_steps(t2, a2, s2, max(args.warmup // 10, 5), world, gb2, overlap=not args.sync_gather)
            res.update(plus_depth_value=n_gpus * N * s2 / dt2, plus_depth_ms_per_step=1e3 * dt2 / s2)
            gb2.close()
    except Exception as e:  # noqa: BLE001  (reported, the process-group numbers stand)
        res["error"] = f"{type(e).__name__}: {e}"
        res["failed_in"] = stage
    # a rank that failed alone leaves the others inside a collective: the watchdog stays armed until every rank
    # has reached this barrier, and all ranks agree on whether the numbers of this leg count
    # 0 = fine, 1 = the exchange could not be set up (every rank, together, nothing left behind), 2 = anything else
    code = torch.tensor([0 if "error" not in res else (1 if stage == "setup" else 2)], device=device, dtype=torch.int32)
    dist.all_reduce(code, op=dist.ReduceOp.MAX)
    worst = int(code.item())
    all_ok = worst == 0
    dog.cancel()
    if not all_ok and "error" not in res:
        res["error"] = "another rank failed"
    report[backend] = res
    if worst == 1:  # e.g. peer mappings refused on this platform: the next backend can still be measured
        return "setup_failed"
    if not all_ok:
        # whatever is left of this leg (a communicator some rank never joined, a device-side wait nobody will
        # release) must not get a chance to block a destructor before rank 0 has printed: keep it alive and let
        # main() leave through os._exit right after the line
        _ABANDONED.append(dict(locals()))
        return False
    how = {"peer_push": "peer push (rows stored into the peers' receive buffers over xGMI, no collective kernel per step)",
           "rccl_thread": "1 RCCL all_gather/step enqueued by the library's worker thread"}[backend]
    if "value" in res and res["value"] > out["value"]:
        out["value"], out["ms_per_step"] = res["value"], res["ms_per_step"]
        out["config"]["sharding"] = f"envs x{n_gpus}, [N, obs_dim+3] rows per step: {how}, " + ("synchronous" if args.sync_gather else "overlapped with the next step")
        out["config"]["exchange_backend"] = backend
    if "plus_depth_value" in res and "plus_depth" in out and res["plus_depth_value"] > out["plus_depth"]["value"]:
        out["plus_depth"]["value"], out["plus_depth"]["ms_per_step"] = res["plus_depth_value"], res["plus_depth_ms_per_step"]
        out["plus_depth"]["exchange"] = backend
    return True


_ABANDONED = []
_JSON_FD = [None]


LINE_LIMIT = 6144  # bytes of the ONE stdout line; the driver keeps ~9 KB of stdout and parses the line from it (r05: a 23 KB line -> "parsed": null)
DETAIL_NAME = "bench_detail.json"


def _r(x, sig=6):
    """floats to `sig` significant digits (the line is a record, not an archive: bench_detail.json keeps full precision)"""
    if isinstance(x, float):
        return float(f"{x:.{sig}g}")
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _roofline_brief(blk):
    """the contract's roofline object and nothing nested: bound / achieved / peak / unit / frac / traffic + the kernel it is
    about, its live launch time, the algorithmic bytes it is priced with and the traffic ratio"""
    if not isinstance(blk, dict):
        return blk
    if "error" in blk:
        return {"error": str(blk["error"])[:160]}
    hbm = blk.get("hbm") or blk  # a kernel labelled bound "valu" keeps its byte view under "hbm": the LINE carries the byte view
    out = {"bound": "hbm", "achieved": hbm.get("achieved"), "peak": hbm.get("peak"), "unit": hbm.get("unit"), "frac": hbm.get("frac"),
           "traffic": hbm.get("traffic"), "kernel": str(blk.get("kernel", ""))[:96], "launch_us": blk.get("launch_us"),
           "algorithmic_bytes_per_launch": hbm.get("algorithmic_bytes_per_launch"),
           "traffic_over_algorithmic": hbm.get("traffic_over_algorithmic"), "traffic_stale": hbm.get("traffic_stale")}
    if blk.get("bound") == "valu":
        out["limited_by"] = "valu issue"
        out["valu_issue_frac_of_guide_peak"] = blk.get("frac")
    if "num_envs" in blk:
        out["num_envs"] = blk["num_envs"]
    return out


def _sensor_brief(leg):
    if not isinstance(leg, dict):
        return leg
    out = _pick(leg, ("value", "unit", "ms_per_step", "steps", "workload", "value_synchronised_episodes", "raycast_launch_us", "exchange"))
    if "workload" in out:
        out["workload"] = str(out["workload"])[:200]
    rr = leg.get("raycast_roofline")
    if isinstance(rr, dict):
        hbm = rr.get("hbm") or rr
        out["raycast_hbm_frac"] = hbm.get("frac")
        out["raycast_traffic_over_algorithmic"] = hbm.get("traffic_over_algorithmic")
    cb = leg.get("cpu_baseline_raycast")
    if isinstance(cb, dict):
        out["cpu_baseline_raycast"] = _pick(cb, ("value", "unit", "cores", "kind"))
    return out


def compact_line(out, detail_path=None):
    """What goes on stdout: the contract's keys, the two objects it asks for (`roofline`, `cpu_baseline`), one number per
    secondary leg, and the path of the file that holds everything else.  Everything measured stays in `out` -> bench_detail.json."""
    if "metric" not in out:  # self-test verdicts and the like: small by construction
        return out
    line = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                       "dtype", "data"))
    cfg = dict(out.get("config", {}))
    for k in ("workload", "sharding"):
        if k in cfg:
            cfg[k] = str(cfg[k])[:220]
    line["config"] = cfg
    if "roofline" in out:
        line["roofline"] = _roofline_brief(out["roofline"])
    if "cpu_baseline" in out:
        cb = _pick(out["cpu_baseline"], ("value", "unit", "cores", "kind", "sample", "measured_here"))
        cb["sample"] = str(cb.get("sample", ""))[:200]
        line["cpu_baseline"] = cb
    for k in ("cpu_baseline_reference", "cpu_baseline_port"):
        if isinstance(out.get(k), dict):
            line[k] = _pick(out[k], ("value", "unit", "cores", "kind", "measured_here", "source"))
    tr = out.get("timed_regions")
    if isinstance(tr, dict):
        line["timed_regions"] = _pick(tr, ("count", "steps_each", "value_is", "value_min", "value_max"))
    line["value_is"] = "median region; de-synchronised episodes (resets in every timed step)"
    for k in ("value_synchronised_episodes", "value_strict_rng"):
        if k in out:
            line[k] = out[k]
    if isinstance(out.get("host"), dict):
        line["host_share_of_step"] = out["host"].get("share_of_step")
    for k in ("roofline_reset_obs", "roofline_at_scale", "roofline_at_scale_lean"):
        if k in out:
            b = _roofline_brief(out[k])
            line[k] = _pick(b, ("kernel", "launch_us", "frac", "traffic_over_algorithmic", "num_envs", "error")) if isinstance(b, dict) else b
    if isinstance(out.get("roofline_step"), dict):
        line["roofline_step"] = _pick(out["roofline_step"], ("launches", "kernel_us_sum", "algorithmic_bytes_per_step", "frac_over_step_time"))
    for k in ("plus_depth", "plus_lidar"):
        if k in out:
            line[k] = _sensor_brief(out[k])
    ex = out.get("exchange")
    if isinstance(ex, dict):
        brief = _pick(ex, ("ms_per_step_with_exchange", "ms_per_step_without_exchange", "all_gather_us_synchronised", "bytes_per_rank",
                           "communicator_ranks", "backend_of_first_leg"))
        if isinstance(ex.get("preflight"), list):
            brief["preflight"] = [{k: (str(v)[:200] if isinstance(v, (dict, str)) else v) for k, v in p.items()} for p in ex["preflight"][:4]]
        for b in ("process_group", "peer_push", "rccl_thread"):
            if isinstance(ex.get(b), dict):
                brief[b] = _pick(ex[b], ("value", "ms_per_step", "plus_depth_value", "ranks_seen", "failed_in"))
                if "error" in ex[b]:
                    brief[b]["error"] = str(ex[b]["error"])[:160]
        line["exchange"] = brief
    par = out.get("parity")
    if isinstance(par, dict):
        line["parity"] = ({"error": str(par["error"])[:160]} if "error" in par else
                          {"max_err_vs_reference": par.get("max_err_vs_reference"),
                           "bit_exact_vs_reference_cr": par.get("bit_exact_vs_reference_with_correctly_rounded_functions")})
    if isinstance(out.get("strict_rng"), dict) and "error" in out["strict_rng"]:
        line["strict_rng_error"] = str(out["strict_rng"]["error"])[:160]
    line.update(_pick(out, ("build_id", "binary_matches_sources")))
    if "error" in out:  # (a run that could not be timed: the preflight verdict line)
        line["error"] = str(out["error"])[:300]
    if detail_path:
        line["detail"] = detail_path
    line = _r(line)
    # the guard: whatever a future key adds, the line that reaches stdout parses from the driver's tail.  Secondary objects go
    # first (they are all in the detail file), the contract's keys never
    for k in ("parity", "exchange", "roofline_step", "roofline_reset_obs", "roofline_at_scale_lean", "roofline_at_scale", "cpu_baseline_port",
              "cpu_baseline_reference", "plus_lidar", "plus_depth", "timed_regions"):
        if len(json.dumps(line)) < LINE_LIMIT:
            break
        if k in line:
            line.pop(k)
            line.setdefault("dropped_for_size", []).append(k)
    assert len(json.dumps(line)) < LINE_LIMIT, "bench line over the size the driver parses"
    return line


def write_detail(out):
    """everything measured, at full precision: bench_detail.json next to bench.py (and under gpurun_out/ when that exists, so
    that a gpurun call brings it back).  Returns the path named in the line (relative to the repo root)."""
    text = json.dumps(out, indent=1)
    path = None
    for d in (os.path.join(ROOT, "gpurun_out"), ROOT):
        if d != ROOT and not os.path.isdir(d):
            continue
        try:
            with open(os.path.join(d, DETAIL_NAME), "w") as f:
                f.write(text)
            path = path or os.path.relpath(os.path.join(d, DETAIL_NAME), ROOT)
        except OSError:
            continue
    return path


def emit_line(out, fd=None):
    """the ONE JSON line, on the process's original stdout: the compact form (< LINE_LIMIT bytes); the full record goes to
    bench_detail.json"""
    fd = fd if fd is not None else _JSON_FD[0]
    detail = write_detail(out) if "metric" in out else None
    line = json.dumps(compact_line(out, detail)) + "\n"
    if fd is None:
        sys.stdout.write(line)
        sys.stdout.flush()
    else:
        os.write(fd, line.encode())


def usable_cpus():
    """hardware threads this process may really use: the affinity mask, capped by the cgroup's CPU quota (a GPU box reports 256
    logical CPUs to os.cpu_count() whatever the container is allowed: 256 OpenMP threads on a few cores ran the port 10x slower
    in one round-6 run than in the next)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def set_omp_threads(n):
    """thread count of the oracle's OpenMP loops from now on (libgomp's omp_set_num_threads); False if it cannot be set"""
    import ctypes as C

    try:
        C.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
        return True
    except OSError:
        return False


def cpu_baseline_dynamics(num_envs, budget_s=12.0):
    """The CPU oracle (a C port of the reference's per-env step, oracle/) timed on this box's
    host cores on the same workload: 8192 envs, position task, Lee position control, k=1."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle as orc
    from oracle_env import OraclePositionEnv

    from aerial_gym_simulator_amd.config.controller_config import lee_controller_config as L
    from aerial_gym_simulator_amd.config.robot_config import BaseQuadCfg
    from aerial_gym_simulator_amd.config.sim_config import BaseSimConfig
    from aerial_gym_simulator_amd.robots.robot_model import robot_params_dict

    pd = robot_params_dict(BaseQuadCfg, L, "position", BaseSimConfig)
    n = num_envs
    mid = lambda a, b: ((np.array(a) + np.array(b)) / 2).astype(np.float32)  # noqa: E731
    gains = [np.tile(mid(getattr(L, f"K_{k}_tensor_max"), getattr(L, f"K_{k}_tensor_min")), (n, 1)) for k in ("pos", "vel", "rot", "angvel")]
    ranges = dict(tau_inc=(0.04, 0.04), tau_dec=(0.04, 0.04), thrust=(0.0, 2.0), kT=(0.00000926312, 0.00001826312))
    env = OraclePositionEnv(pd, n, 500, gains, BaseQuadCfg.init_config.min_init_state, BaseQuadCfg.init_config.max_init_state, ranges)
    rng = np.random.default_rng(0)
    draws = lambda: (rng.random((n, 13), np.float32), rng.random((n, 4), np.float32), rng.random((n, 4), np.float32),  # noqa: E731
                     rng.random((n, 4), np.float32), rng.random((n, 4), np.float32))
    env.reset_masked(np.ones(n, np.uint8), *draws())
    actions = [rng.uniform(-1, 1, (n, 4)).astype(np.float32) for _ in range(8)]
    d = draws()
    for i in range(5):
        env.step(actions[i % 8], d)
    # the thread count that is fastest HERE: all usable CPUs, or fewer where hyper-threads / a quota make that slower (1 s each)
    usable = usable_cpus()
    cores, tried = int(os.environ.get("OMP_NUM_THREADS", usable)), {}
    if "OMP_NUM_THREADS" not in os.environ:
        for cand in sorted({usable, max(1, usable // 2), min(usable, 32), min(usable, 8)}, reverse=True):
            if not set_omp_threads(cand):
                break
            env.step(actions[0], d)
            n_probe, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < 1.0:
                env.step(actions[n_probe % 8], d)
                n_probe += 1
            tried[cand] = n_probe / (time.perf_counter() - t0)
        if tried:
            cores = max(tried, key=tried.get)
            set_omp_threads(cores)
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s - len(tried):
        for i in range(20):
            env.step(actions[i % 8], d)
        steps += 20
    dt = time.perf_counter() - t0
    return {
        "value": n * steps / dt,
        "unit": "env-steps/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{steps} env steps of {n} envs ({dt:.1f} s): oracle C port of the per-env step (OpenMP over envs, {cores} threads: "
                  f"the fastest of {sorted(tried)} tried for 1 s each; os.cpu_count() = {os.cpu_count()}), same task/config as the GPU run",
        "threads_tried_steps_per_s": {str(k): round(v, 1) for k, v in tried.items()},
    }


def cpu_baseline_raycast(task, budget_s=8.0, sample_envs=512):
    """The CPU oracle's ray-cast (C port of the reference's camera kernel over a median-split BVH, OpenMP over
    envs) on a sample of the GPU task's own scenes and sensor poses: frames of `sample_envs` envs for ~budget_s."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle as orc

    env = task.sim_env
    sc, sen = env.scene, env.robot_manager.warp_sensor
    m = min(sample_envs, env.num_envs)
    npy = lambda t: np.ascontiguousarray(t[:m].detach().cpu().numpy())  # noqa: E731
    tris, seg, pos, quat = npy(sc.tri_world), npy(sc.tri_seg), npy(sen.sensor_position), npy(sen.sensor_orientation)
    cfg = sen.cfg
    kinv, cx, cy = orc.camera_kinv(cfg.width, cfg.height, cfg.horizontal_fov_deg)
    cores = usable_cpus()
    if "OMP_NUM_THREADS" not in os.environ:
        set_omp_threads(cores)
    orc.raycast_camera(cfg.width, cfg.height, kinv, cfg.max_range, cx, cy, "depth", pos, quat, tris, seg, use_bvh=True)  # warm-up
    frames, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        orc.raycast_camera(cfg.width, cfg.height, kinv, cfg.max_range, cx, cy, "depth", pos, quat, tris, seg, use_bvh=True)
        frames += 1
    dt = time.perf_counter() - t0
    rays = frames * m * cfg.num_sensors * cfg.width * cfg.height
    return {"value": frames * m / dt, "unit": "env-frames/s", "rays_per_s": rays / dt, "cores": int(os.environ.get("OMP_NUM_THREADS", 0)) or cores,
            "kind": "port", "sample": f"{frames} frames of {m} envs ({dt:.1f} s): oracle C port of the depth+seg camera kernel, BVH build "
                                      "included in every frame (OpenMP over envs), scenes and poses of the GPU run"}


def sensor_leg(args, workload, num_envs, device, rank, world, use_dist, primary_backend, copy_gbs, label, cpu_baseline=True):
    """One sensor configuration of BASELINE.json (configs[2] camera, configs[3] LiDAR) timed like `value`: a short region with
    the episodes as reset() leaves them, then de-synchronised episodes (`value`), median of up to 3 regions; on one GPU also
    the ray-cast launch on its own with its roofline object."""
    from aerial_gym_simulator_amd.sharding import StepGather

    n_gpus = max(world, 1)
    t2 = make_task(workload, num_envs, device, args.strict_rng, rank)
    t2.reset()
    N, A = t2.num_envs, t2.task_config.action_space_dim
    g = torch.Generator(device=device).manual_seed(4321 + rank)
    a2 = [torch.rand(N, A, device=device, generator=g) * 2 - 1 for _ in range(4)]
    s2 = min(max(args.steps // 10, 20), 300)
    gb2 = None
    if use_dist:
        gb2 = StepGather(N, t2.task_obs["observations"].shape[1], device, env=t2.sim_env, reward=t2.rewards, backend=primary_backend)
    w2 = max(args.warmup // 10, 5)
    ep = int(t2.sim_env.global_tensor_dict["episode_count"].sum().item())
    dt_sync = timed_steps(t2, a2, s2, w2, world, gb2, overlap=not args.sync_gather)
    sync_resets, ep = resets_per_step(t2, ep, w2 + s2)
    desync = desynchronise_episodes(t2)
    regions = max(1, min(args.regions, 3))
    dts2 = sorted(timed_steps(t2, a2, s2, w2 if r == 0 else 0, world, gb2, overlap=not args.sync_gather) for r in range(regions))
    desync["measured_resets_per_step"], ep = resets_per_step(t2, ep, w2 + regions * s2)
    dt2 = dts2[len(dts2) // 2]
    leg = None
    if rank == 0:
        cfgs = t2.sim_env.robot_manager.warp_sensor.cfg
        rays = N * cfgs.num_sensors * cfgs.height * cfgs.width
        leg = {"value": n_gpus * N * s2 / dt2, "unit": "env-steps/s", "n_gpus": n_gpus, "steps": s2, "num_envs_per_gpu": N,
               "ms_per_step": 1e3 * dt2 / s2, "timed_regions_ms_per_step": [1e3 * x / s2 for x in dts2],
               "episodes": dict(desync, state="de-synchronised (steady state)"),
               "value_synchronised_episodes": n_gpus * N * s2 / dt_sync,
               "synchronised_episodes": {"ms_per_step": 1e3 * dt_sync / s2, "measured_resets_per_step": sync_resets,
                                         "note": "episodes as reset() leaves them (crashes under random actions still reset envs)"},
               "workload": label}
    if rank == 0 and world == 1:
        kt2 = kernel_time_raycast(t2)
        per_env = raycast_bytes_per_env(t2)
        leg.update({
            "raycast_launch_us": kt2 * 1e6, "rays_per_s_kernel": rays / kt2,
            "raycast_roofline": roofline_block("k_raycast (one frame, all envs)", kt2, per_env * N, raycast_key(t2), copy_gbs, bound="valu",
                                               note="packet traversal: bound by vector-instruction issue and by the latency of its dependent "
                                                    "node fetches (profiles/r04_raycast_variants.txt), not by HBM bytes")})
        if cpu_baseline and not args.no_cpu_baseline:
            leg["cpu_baseline_raycast"] = cpu_baseline_raycast(t2)
            leg["gpu_frames_per_s_kernel"] = N / kt2
    if gb2 is not None:
        gb2.close()
    del t2, gb2
    torch.cuda.empty_cache()
    return leg


def exchange_selftest(args, world, rank, device, limit_s=60.0, preflight=None):
    """`--exchange-selftest-only`: first contact with an N-GPU node made cheap.  Per backend: the communicator / IPC mappings are
    built, `--selftest-steps` real env steps run with the per-step exchange, every rank checksums the rows it SENT and every
    slice it RECEIVED (bit patterns summed as integers: exact, order-independent), the sent checksums travel over the process
    group and are compared step by step -- the checks of tests/exchange_world2_worker.py.  One JSON line per backend on rank 0's
    stdout: which leg failed, in which stage, and which (receiver, sender) pair first disagreed at which step.  Nothing is timed
    for the record; a leg that does not come back within `limit_s` prints what it has and ends the process.
    `preflight` (a list): the same checks as the FIRST thing a timed N > 1 run does (round 6: whatever command a multi-GPU lease
    runs first, the verdicts come first) -- `--preflight-steps` steps per backend, verdicts on stderr and appended to the list
    (they end up in the line's `exchange.preflight`); a leg that hangs still ends the process, with a parseable line on stdout
    that names it; a leg that fails is reported and the timed legs run anyway (each is watched by its own watchdog)."""
    import threading

    import torch.distributed as dist
    from aerial_gym_simulator_amd.sharding import StepGather

    steps, N = int(args.preflight_steps if preflight is not None else args.selftest_steps), args.num_envs

    def emit(res):
        if preflight is None:
            emit_line(res)
        else:
            print("[bench preflight] " + json.dumps(res)[:1500], file=sys.stderr, flush=True)

    for backend in ("process_group", "peer_push", "rccl_thread"):
        res = {"selftest": "exchange", "backend": backend, "world": world, "steps": steps, "num_envs_per_rank": N, "ok": False}
        stage = ["build"]

        def give_up(res=res, stage=stage):
            res["error"] = f"no completion within {limit_s:.0f} s (rank {rank} was in stage '{stage[0]}')"
            if rank == 0:
                # (preflight: the ONE stdout line of the run is this verdict -- the timed legs never started)
                emit_line(res if preflight is None else {"metric": "env-steps/sec (not measured: the exchange preflight hung)", "value": None,
                                                         "n_gpus": world, "error": res["error"], "exchange": {"preflight": preflight + [res]}})
            os._exit(3)

        dog = threading.Timer(limit_s, give_up)
        dog.daemon = True
        dog.start()
        t0 = time.perf_counter()
        mine_bad = {}
        try:
            task = make_task("dynamics", N, device, False, rank)
            task.reset()
            desynchronise_episodes(task)  # resets (and the rows they rewrite) are part of every step
            d = task.task_obs["observations"].shape[1]
            g = torch.Generator(device=device).manual_seed(99 + rank)
            actions = [torch.rand(N, task.task_config.action_space_dim, device=device, generator=g) * 2 - 1 for _ in range(8)]
            stage[0] = "setup"
            sg = StepGather(N, d, device, env=task.sim_env, reward=task.rewards, backend=backend)
            stage[0] = "run"
            res["communicator"] = dict(zip(("rank", "ranks"), sg.comm_info()))
            if backend == "peer_push":
                res["connection_selftest"], res["kernel_push"] = sg.push_selftest, bool(sg._kernel_push)
            if res["communicator"]["ranks"] != world:
                raise RuntimeError(f"the communicator spans {res['communicator']['ranks']} ranks, the job has {world}")
            W, lag = d + 3, sg.lag
            own = torch.zeros(steps, dtype=torch.int64, device=device)
            got = torch.zeros(steps, world, dtype=torch.int64, device=device)
            own_ok = torch.ones((), dtype=torch.bool, device=device)
            history = []
            for t in range(steps):
                obs, rew, term, trunc, _ = task.step(actions[t % 8])
                out = sg.exchange(task.sim_env._parity, overlap=True)
                rows = torch.cat([obs["observations"], rew[:, None], term[:, None].float(), trunc[:, None].float()], dim=1)
                own[t] = rows.view(torch.int32).long().sum()
                history.append(rows)
                if out is not None:
                    got[t - lag] = out.view(torch.int32).view(world, N, W).long().sum(dim=(1, 2))
                    own_ok &= torch.equal(out.view(world, N, W)[rank], history[-lag - 1])
                history = history[-3:]
            sg.flush()
            torch.cuda.synchronize()
            stage[0] = "compare"
            sent = torch.zeros(world * steps, dtype=torch.int64, device=device)
            dist.all_gather_into_tensor(sent, own)
            last = steps - lag
            wrong = got[:last] != sent.view(world, steps).t()[:last]
            for sender in range(world):
                idx = wrong[:, sender].nonzero()
                if idx.numel():
                    mine_bad[sender] = int(idx[0])
            res_rank = {"rank": rank, "ok": not mine_bad and bool(own_ok), "own_slice_ok": bool(own_ok),
                        "first_bad_step_by_sender": mine_bad, "resets": int(task.sim_env.global_tensor_dict["episode_count"].sum()) - N}
            stage[0] = "close"
            sg.close()
            del task, sg
            torch.cuda.empty_cache()
        except Exception as e:  # noqa: BLE001
            res_rank = {"rank": rank, "ok": False, "error": f"{type(e).__name__}: {e}", "failed_in": stage[0]}
        stage[0] = "verdict"
        per_rank = [None] * world
        dist.all_gather_object(per_rank, res_rank)
        dog.cancel()
        res.update(ok=all(r["ok"] for r in per_rank), per_rank=per_rank, seconds=time.perf_counter() - t0)
        if rank == 0:
            emit(res)
        if preflight is not None:
            bad = [r for r in per_rank if not r["ok"]]
            preflight.append({"backend": backend, "ok": res["ok"], "steps": steps, "seconds": round(res["seconds"], 2),
                              "first_failure": ({k: bad[0].get(k) for k in ("rank", "error", "failed_in", "first_bad_step_by_sender")} if bad else None)})
        if any("error" in r and r.get("failed_in") not in ("setup",) for r in per_rank):
            # something half-built may be left behind (a communicator some rank never joined): no further legs, no teardown
            if preflight is not None and rank == 0:
                emit_line({"metric": "env-steps/sec (not measured: the exchange preflight failed)", "value": None, "n_gpus": world,
                           "error": f"exchange preflight: backend {backend} failed", "exchange": {"preflight": preflight}})
            sys.stdout.flush()
            os._exit(0 if rank != 0 else 4)


def ensure_ranks(args):
    """`--gpus N` is a request for N ranks, one per GPU, and it is honoured or refused -- never silently reduced:

    * WORLD_SIZE set (the process was started by torch.distributed.run / torchrun): it must equal N;
    * WORLD_SIZE unset and N > 1: this process becomes the launcher -- it re-runs itself under
      `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>`
      with the same arguments and returns the children's exit code (rank 0 prints the ONE JSON line on the inherited stdout);
    * fewer than N HIP devices visible: exit non-zero naming both numbers, before anything is launched.
    AGX_BENCH_SPAWN=1 sends N = 1 through the same launcher (tests)."""
    env_world = os.environ.get("WORLD_SIZE")
    visible = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if env_world is not None:
        if int(env_world) != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={env_world}: start {args.gpus} ranks "
                             f"(--nproc-per-node {args.gpus}) or pass --gpus {env_world}")
    if visible < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} requested but {visible} HIP device(s) visible "
                         f"(HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES', '<unset>')}, "
                         f"ROCR_VISIBLE_DEVICES={os.environ.get('ROCR_VISIBLE_DEVICES', '<unset>')}); there is no CPU fallback and no "
                         "smaller job is run in its place")
    if env_world is not None:
        return
    if args.gpus == 1 and os.environ.get("AGX_BENCH_SPAWN") != "1":
        return
    import socket
    import subprocess

    import random

    port = None
    for _ in range(200):  # below the kernel's ephemeral range: a port from bind(0) can be taken again before the launcher listens on it
        cand = random.randrange(15000, 30000)
        with socket.socket() as s:
            try:
                s.bind(("127.0.0.1", cand))
            except OSError:
                continue
        port = cand
        break
    if port is None:
        raise SystemExit("bench.py: no free rendezvous port on 127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {args.gpus} without a launcher: starting {args.gpus} rank(s): {' '.join(cmd)}", file=sys.stderr, flush=True)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL and the peer-push mappings need it on this driver)
    env.pop("AGX_BENCH_SPAWN", None)
    raise SystemExit(subprocess.call(cmd, env=env))


def rank_diagnostics(rank, world, local_rank):
    """What this rank sees before anything is timed, on stderr: the first contact with an N-GPU node explains itself."""
    n = torch.cuda.device_count()
    row = []
    for j in range(n):
        try:
            row.append(1 if j == local_rank else int(torch.cuda.can_device_access_peer(local_rank, j)))  # hipDeviceCanAccessPeer
        except Exception:  # noqa: BLE001
            row.append(-1)
    info = {"rank": rank, "world": world, "local_rank": local_rank, "devices_visible": n, "device": torch.cuda.get_device_name(local_rank),
            "can_access_peer": row, "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
            "HIP_VISIBLE_DEVICES": os.environ.get("HIP_VISIBLE_DEVICES")}
    print("[bench rank %d/%d] %s" % (rank, world, json.dumps(info)), file=sys.stderr, flush=True)
    return info


def main():
    args = parse()
    ensure_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP GPU (no CPU fallback); use gpurun")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} has LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} HIP device(s) are visible")
    torch.cuda.set_device(local_rank)
    diag = rank_diagnostics(rank, world, local_rank)
    device = f"cuda:{local_rank}"
    use_dist = world > 1 or os.environ.get("AGX_BENCH_FORCE_DIST") == "1" or args.exchange_selftest_only
    json_fd = None
    if use_dist:
        # RCCL prints a version banner on stdout when a communicator goes away: keep stdout for the ONE JSON line
        sys.stdout.flush()
        json_fd = _JSON_FD[0] = os.dup(1)
        os.dup2(2, 1)
    if use_dist:
        import torch.distributed as dist

        if world == 1:  # debugging aid: exercise the collective path on a single GPU
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            dist.init_process_group("nccl", rank=0, world_size=1)
        else:
            dist.init_process_group("nccl")  # RCCL over xGMI; rank / world size / master from the env
    n_gpus = max(world, 1)
    if args.exchange_selftest_only:
        import torch.distributed as dist

        exchange_selftest(args, world, rank, device)
        dist.barrier()
        dist.destroy_process_group()
        return
    preflight = None
    if use_dist and args.preflight_steps > 0 and (world > 1 or os.environ.get("AGX_BENCH_PREFLIGHT") == "1"):
        # first contact with an N-GPU node: which backend, which stage, which rank pair -- before the first timed step
        preflight = []
        exchange_selftest(args, world, rank, device, limit_s=180.0, preflight=preflight)  # (a first RCCL communicator can take tens of seconds)
        import torch.distributed as dist

        dist.barrier()
    task = make_task(args.workload, args.num_envs, device, args.strict_rng, rank)
    task.reset()
    N, A = task.num_envs, task.task_config.action_space_dim
    g = torch.Generator(device=device).manual_seed(1234 + rank)
    actions = [torch.rand(N, A, device=device, generator=g) * 2 - 1 for _ in range(16)]
    gather_buf = None
    from aerial_gym_simulator_amd.sharding import StepGather

    primary_backend = args.exchange if args.exchange in ("rccl_thread", "peer_push") else "process_group"
    if use_dist:
        gather_buf = StepGather(N, task.task_obs["observations"].shape[1], device, env=task.sim_env, reward=task.rewards,
                                backend=primary_backend)
    if gather_buf is not None:
        comm_rank, comm_ranks = gather_buf.comm_info()
        print(f"[bench rank {rank}/{world}] first leg: exchange backend {gather_buf.backend}, communicator reports rank {comm_rank} of {comm_ranks}",
              file=sys.stderr, flush=True)
        if comm_ranks != max(world, 1) or comm_rank != rank:
            raise SystemExit(f"the step exchange's communicator reports rank {comm_rank} of {comm_ranks}; the job is rank {rank} of {world}")
    # Right after reset() every episode is at step 0: a region shorter than episode_len contains no reset at all.  That number is
    # measured first and kept as `value_synchronised_episodes` (rounds 1-4 reported it as `value` on short runs); then the episodes
    # are spread uniformly over their length -- the steady state of a long-running job -- and THAT is what `value` times.
    sync_steps = max(1, min(args.steps, 100))
    sync_warm = min(args.warmup, 50)
    ep0 = int(task.sim_env.global_tensor_dict["episode_count"].sum().item())
    dts_sync = sorted(timed_steps(task, actions, sync_steps, sync_warm if r == 0 else 0, world, gather_buf, overlap=not args.sync_gather)
                      for r in range(3))
    sync_resets, ep0 = resets_per_step(task, ep0, sync_warm + 3 * sync_steps)
    desync = desynchronise_episodes(task)
    # `value` = the MEDIAN of `--regions` timed regions of exactly `--steps` steps each (each bracketed by barrier + device
    # synchronize, maximum over ranks); minimum and maximum are reported next to it.  One region of the driver's 20 steps lasts
    # 0.25 ms: a single scheduler hiccup is +-30 % of it.
    dts, hosts = [], []
    for r in range(max(1, args.regions)):
        dts.append(timed_steps(task, actions, args.steps, args.warmup if r == 0 else 0, world, gather_buf, overlap=not args.sync_gather))
        hosts.append(timed_steps.last_host_s)
    desync["measured_resets_per_step"], ep0 = resets_per_step(task, ep0, args.warmup + len(dts) * args.steps)
    dts_sorted = sorted(dts)
    dt = dts_sorted[len(dts_sorted) // 2]
    value = n_gpus * N * args.steps / dt
    exchange = exchange_diagnostics(task, actions, args, world, gather_buf, dt) if use_dist else None
    out = {
        "metric": f"env-steps/sec at N_envs={N} per GPU (" + ("dynamics-only" if args.workload == "dynamics" else "+" + args.workload + " sensor") + ")",
        "value": value,
        "unit": "env-steps/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "timed_regions": {"count": len(dts), "steps_each": args.steps, "value_is": "median",
                          "ms_per_step": [1e3 * x / args.steps for x in dts],
                          "value_min": n_gpus * N * args.steps / dts_sorted[-1], "value_max": n_gpus * N * args.steps / dts_sorted[0]},
        "value_is": "steady state with de-synchronised episodes: num_envs / episode_len envs reset in EVERY timed step (since round 5; rounds 1-4 "
                    "timed short regions right after reset(), where no env resets: that figure is `value_synchronised_episodes`)",
        "episodes": dict(desync, state="de-synchronised (steady state: resets in every step)"),
        "value_synchronised_episodes": n_gpus * N * sync_steps / dts_sync[1],
        "synchronised_episodes": {"ms_per_step": [1e3 * x / sync_steps for x in dts_sync], "steps_each": sync_steps, "regions": 3,
                                  "measured_resets_per_step": sync_resets,
                                  "note": "every episode at the same step (right after reset()): no env truncates inside these regions -- "
                                          "the reset branch of the step's second launch does no work; what rounds 1-4 reported for short runs"},
        "host": {"enqueue_us_per_step": sorted(1e6 * h / args.steps for h in hosts)[len(hosts) // 2],
                 "share_of_step": sorted(h / d for h, d in zip(hosts, dts))[len(hosts) // 2],
                 "note": "wall time of the Python loop that enqueues a region's launches (before the closing synchronize) over the region's "
                         "wall time: 1.0 = the host is the bound, the device idles between steps"},
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": {"dynamics": "base_quadrotor position_setpoint_task, lee_position_control, empty_env, 1 sub-step/step (BASELINE configs[1])",
                         "depth": "base_quadrotor navigation_task, lee_velocity_control, 100 random boxes + 6 walls, 10 sub-steps/step, 64x48 depth+seg camera (BASELINE configs[2])",
                         "lidar": "base_octarotor navigation_task, rov_fully_actuated_control (7-D position + attitude command), 100 boxes + 6 walls, 10 sub-steps/step, 32x512 LiDAR range+seg (BASELINE configs[3] as written)",
                         "lidar_velocity": "base_octarotor navigation_task, octarotor_velocity_control, 100 boxes + 6 walls, 32x512 LiDAR range+seg",
                         "lidar_nav": "magpie lidar_navigation_task, magpie_acceleration_control, env_with_lidar_nav_obstacles (91 assets), 48x120 world-frame point-cloud LiDAR, 10 sub-steps/step (SURVEY 8 f2)"}[args.workload],
            "num_envs_per_gpu": N,
            "num_envs_total": n_gpus * N,
            "sharding": (f"envs x{n_gpus}, 1 RCCL all_gather/step of [N, obs_dim+3] rows, "
                         + ("synchronous" if args.sync_gather else "overlapped with the next step")
                         + f", enqueued by {primary_backend}") if use_dist else "single GPU",
            "rng": "strict (reference torch stream, host sync/step)" if args.strict_rng else "sync-free (device Philox4x32-10)",
        },
    }
    copy_gbs = hbm_copy_gbs(device) if rank == 0 else None
    if rank == 0 and args.workload == "dynamics":
        timing, k = kernel_time_dynamics(task, actions)
        post_timing = timing.pop("_post_step", None)
        ekey = env_step_key(task, k)
        out["roofline"] = roofline_block(
            ekey.rsplit("_", 1)[0] + " (sub-step(s) + reward epilogue)", timing["primary"], BYTES_DYNAMICS_KERNEL * k * N, ekey, copy_gbs, timing,
            note="8192 envs = 512 one-wave workgroups on 1024 SIMDs, 1.3 MB per launch: neither HBM nor the vector ALUs can be filled, the "
                 "launch is latency bound (one wave's instruction stream + two memory round trips); roofline_at_scale prices the "
                 "one-lane-per-env kernel at 2^21 envs")
    if rank == 0 and args.workload == "dynamics" and post_timing:
        post = post_timing
        out["roofline_reset_obs"] = roofline_block(
            "k_reset_masked_quad_obs (reset of the flagged envs + observation: the step's second launch)", post["primary"],
            BYTES_RESET_OBS_KERNEL * N, "k_reset_masked_quad_obs_%d" % (4 * N), copy_gbs, post,
            note="per env: 52 B state + 24 B body-frame velocities + 12 B target + 3 flag bytes read, 52 B observation written "
                 "(+ a 64 B exchange row per destination when a step exchange is bound: not in this run); the reset branch moves "
                 "nothing on a step without resets.  Latency bound like the env-step launch (512 waves, one dependent chain).")
        out["roofline_step"] = {"launches": 2, "kernel_us_sum": (timing["primary"] + post["primary"]) * 1e6,
                                "share_env_step": timing["primary"] / (timing["primary"] + post["primary"]),
                                "share_reset_obs": post["primary"] / (timing["primary"] + post["primary"]),
                                "algorithmic_bytes_per_step": BYTES_ENV_STEP * N,
                                "achieved_gbs_over_kernel_time": BYTES_ENV_STEP * N / (timing["primary"] + post["primary"]) / 1e9,
                                "achieved_gbs_over_step_time": BYTES_ENV_STEP * N / (dt / args.steps) / 1e9,
                                "frac_over_step_time": BYTES_ENV_STEP * N / (dt / args.steps) / 1e9 / HBM_PEAK_GBS}
    if exchange is not None:
        exchange["communicator_ranks"] = exchange["ranks_seen"] = gather_buf.comm_info()[1]  # checked against WORLD_SIZE above
        exchange["backend_of_first_leg"] = gather_buf.backend
        exchange["rank0_devices"] = diag
        if preflight is not None:
            exchange["preflight"] = preflight
        out["exchange"] = exchange
    if rank == 0 and args.workload != "dynamics":
        kt = kernel_time_raycast(task)
        cfgs = task.sim_env.robot_manager.warp_sensor.cfg
        out["roofline"] = roofline_block(
            "k_raycast (one frame, all envs)", kt, raycast_bytes_per_env(task) * N, raycast_key(task), copy_gbs, bound="valu",
            note="packet traversal is bound by vector-instruction issue, not by HBM: the scene (127 KB/env) is read once per frame",
            rays_per_s=N * cfgs.num_sensors * cfgs.height * cfgs.width / kt)
    if rank == 0 and args.workload == "dynamics" and world == 1:
        # same kernel where the roofline is meaningful (N = 2^21 envs, 319 MB per launch)
        try:
            big = make_task("dynamics", 1 << 21, device, False, lean=False)
            big.reset()
            gb = torch.Generator(device=device).manual_seed(7)
            ab = [torch.rand(1 << 21, A, device=device, generator=gb) * 2 - 1]
            for _ in range(3):
                big.step(ab[0])
            timing2, k2 = kernel_time_dynamics(big, ab, reps=30)
            timing2.pop("_post_step", None)
            ekey2 = env_step_key(big, k2)
            out["roofline_at_scale"] = roofline_block(
                ekey2.rsplit("_", 1)[0] + " (default: every dict tensor maintained)", timing2["primary"], BYTES_DYNAMICS_KERNEL * k2 * (1 << 21), ekey2, copy_gbs, timing2,
                note="the DEFAULT at every batch size: every tensor the dict exposes maintained every step: the kernel is limited by the HBM "
                     "traffic it really moves (`traffic`: the derived tensors, actions / prev_actions and per-env parameters on top of the "
                     "algorithmic bytes) -- 4.7 TB/s of the 6.3 TB/s a copy kernel reaches",
                num_envs=1 << 21, env_steps_per_s_kernel_only=(1 << 21) / timing2["primary"])
            del big
            torch.cuda.empty_cache()
            # the same kernel with AGX_LAUNCH_LEAN (args={"lean_step": True}): the tensors that exist only to be looked at through
            # the dict are not stored every step (recomputed when a key is read)
            nl = LEAN_AT_SCALE_ENVS
            big = make_task("dynamics", nl, device, False, lean=True)  # explicit opt-in: args={"lean_step": True}
            big.reset()
            ab = [torch.rand(nl, A, device=device, generator=gb) * 2 - 1]
            for _ in range(3):
                big.step(ab[0])
            timing3, k3 = kernel_time_dynamics(big, ab, reps=30)
            timing3.pop("_post_step", None)
            ekey3 = env_step_key(big, k3)
            assert big.sim_env._lean
            out["roofline_at_scale_lean"] = roofline_block(
                ekey3.rsplit("_", 1)[0] + " with AGX_LAUNCH_LEAN (opt-in: args={'lean_step': True})", timing3["primary"],
                BYTES_DYNAMICS_KERNEL * k3 * nl, ekey3, copy_gbs, timing3, num_envs=nl, env_steps_per_s_kernel_only=nl / timing3["primary"],
                reduced_tensor_maintenance=True,
                note="OPT-IN variant (args={'lean_step': True}); same kernel, same results; Euler angles / vehicle-frame tensors / action history are not maintained per step (88 of the "
                     "~330 B an env moves; recomputed when a dict key is read).  Bound by the bytes it really moves (`traffic`) and by "
                     "vector-instruction issue (`valu`); 2 / 3 / 4 waves per SIMD measure within 4 % of each other "
                     "(profiles/r04_at_scale_experiments.txt)")
            del big
        except Exception as e:  # noqa: BLE001
            out.setdefault("roofline_at_scale", {"error": f"{type(e).__name__}: {e}"})
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "dynamics":
        port = cpu_baseline_dynamics(N)
        ref = cpu_baseline_reference()
        if ref is not None and ref.get("measured_here"):  # the reference itself, on this box's cores: THE cpu baseline
            out["cpu_baseline"], out["cpu_baseline_port"] = ref, port
        else:
            out["cpu_baseline"] = port
            if ref is not None:
                out["cpu_baseline_reference"] = ref
    if rank == 0 and world == 1 and args.workload == "dynamics":
        try:
            out["parity"] = live_parity(device)
        except Exception as e:  # noqa: BLE001
            out["parity"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        from aerial_gym_simulator_amd import _build as _b
        from aerial_gym_simulator_amd import _lib as _l

        out["source_hash"] = _b.source_hash()
        out["build_id"] = _l.build_id()  # what the loaded libaerialgym_hip.so says it was built from
        out["binary_matches_sources"] = _l.binary_matches_sources()
    if args.workload == "dynamics":
        del task, gather_buf
        torch.cuda.empty_cache()
    if not args.no_depth and args.workload == "dynamics":
        # the "+depth sensor" half of the metric: BASELINE configs[2] on every GPU (= configs[4] when N > 1:
        # 8192 envs per rank + the per-step all-gather); fewer steps: ~2 ms each
        leg = sensor_leg(args, "depth", args.num_envs, device, rank, world, use_dist, primary_backend, copy_gbs,
                         "navigation_task, 8192 envs per GPU, 64x48 depth+seg camera, 100 boxes + 6 walls, 10 sub-steps/step"
                         + (" (BASELINE configs[4] sharding, 1 all_gather/step)" if world > 1 else " (BASELINE configs[2])"))
        if rank == 0:
            out["plus_depth"] = leg
    if not args.no_lidar and args.workload == "dynamics" and world == 1:
        # BASELINE configs[3]: fully-actuated octarotor + 32 x 512 LiDAR range + segmentation, 4096 envs, one GPU
        leg = sensor_leg(args, "lidar", 4096, device, rank, world, False, primary_backend, copy_gbs,
                         "navigation_task_fully_actuated_lidar, 4096 envs, base_octarotor + rov_fully_actuated_control (7-D command), 32x512 LiDAR "
                         "range+seg, 100 boxes + 6 walls, 10 sub-steps/step (BASELINE configs[3])", cpu_baseline=False)
        if rank == 0:
            out["plus_lidar"] = leg
    if not args.no_strict and args.workload == "dynamics" and world == 1 and not args.strict_rng:
        # the configuration whose parity with the reference's SEEDS is proven (tests: strict traces): every draw through torch's
        # generator with the reference's calls, shapes and order, and -- like the reference's nonzero() -- one host read of the
        # reset flag per step (env_manager.py:364-375, base_multirotor.py:177-205)
        try:
            ts = make_task("dynamics", args.num_envs, device, True, rank)
            ts.reset()
            dstrict = desynchronise_episodes(ts)
            epS = int(ts.sim_env.global_tensor_dict["episode_count"].sum().item())
            ss = min(args.steps, 500)
            dS = sorted(timed_steps(ts, actions, ss, min(args.warmup, 50) if r == 0 else 0, world, None) for r in range(3))
            dstrict["measured_resets_per_step"], _ = resets_per_step(ts, epS, min(args.warmup, 50) + 3 * ss)
            out["value_strict_rng"] = N * ss / dS[1]
            out["strict_rng"] = {"value": N * ss / dS[1], "unit": "env-steps/s", "ms_per_step": 1e3 * dS[1] / ss, "steps": ss,
                                 "timed_regions_ms_per_step": [1e3 * x / ss for x in dS], "episodes": dstrict,
                                 "rng": "strict (reference torch stream: rand_like for all N envs per draw in the reference's order, one host "
                                        "synchronisation per step to learn whether any env resets)",
                                 "workload": "same task / config / sizes as `value`"}
            del ts
        except Exception as e:  # noqa: BLE001
            out["strict_rng"] = {"error": f"{type(e).__name__}: {e}"}
    if use_dist and args.exchange == "auto" and args.workload == "dynamics":
        try:
            del t2, gb2
        except NameError:
            pass
        if library_exchange_leg(args, world, rank, device, out, "peer_push"):  # (True, or "setup_failed": nothing half-built)
            library_exchange_leg(args, world, rank, device, out, "rccl_thread")
    if rank == 0:
        emit_line(out, json_fd)
    if _ABANDONED:  # every rank took the same decision (all-reduced): no teardown of a half-built exchange
        sys.stdout.flush()
        os._exit(0)
    if use_dist:
        import torch.distributed as dist

        dist.barrier()  # rank 0 may still have been timing kernels for the roofline keys
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
