"""CPU: register / scratch budget of the shipped gfx950 kernels, read from the code-object metadata of
libaerialgym_hip.so (no GPU needed).  Guards what DESIGN.md section 3 states about occupancy:

* every k_env_step instance launched for n <= 65536 envs (`WIDE`: one-wave workgroups, compiled for one wave
  per SIMD) has no VGPR spill and not a single scratch instruction;
* the straight-line quadrotor kernels of BASELINE configs 1/2 (`SINGLE`, 256-thread workgroups, 3 waves per
  SIMD) have no spill and no scratch either;
* the ray-cast kernels of configs 3/4 (BASIC / NORMAL, camera and LiDAR) stay at <= 64 VGPRs (8 waves per
  SIMD), the stereo variant at <= 80 (6 waves); none of them uses scratch or static LDS."""
import os
import re
import subprocess

import pytest

import codeobj
from aerial_gym_simulator_amd import _build

pytestmark = pytest.mark.skipif(not codeobj.tools_available(), reason="objcopy / ROCm LLVM tools not found")


@pytest.fixture(scope="module")
def meta():
    assert os.path.exists(_build.LIB_PATH), "build the library first (python -m aerial_gym_simulator_amd._build)"
    return codeobj.kernel_metadata(_build.LIB_PATH)


def _env_step(meta):
    out = {}
    for name, rec in meta.items():
        m = re.match(r"void agx::k_env_step<(\d+), (\d+), (true|false), (true|false)>", name)
        if m:
            out[(int(m.group(1)), int(m.group(2)), m.group(3) == "true", m.group(4) == "true")] = rec
    return out


def test_env_step_instances_and_spills(meta):
    ks = _env_step(meta)
    assert len(ks) == 3 * 9 * 2 * 2  # motors {4,6,8} x 9 controller ids (8 laws + external wrench) x {single, k-loop} x {wide, 256-thread}
    report = []
    for (M, C, single, wide), r in sorted(ks.items()):
        if wide:
            assert r["vgpr_spill_count"] == 0, (M, C, single, r)
            assert r["max_flat_workgroup_size"] == 64
        elif single and M == 4:
            # (a non-zero private segment here is frame slots the SGPR spiller reserved -- the float64 polynomial coefficients
            # live in scalar register pairs -- and did not need: test_wide_env_step_kernels_issue_no_scratch_instruction checks
            # these instances instruction by instruction)
            # (controller id 8 -- the host-evaluated plug-ins, one launch per sub-step with torch code in between -- carries the
            #  external-robot switch AGX_LAUNCH_BODY_WRENCH since round 5: a few more reserved SGPR-spill slots)
            assert r["vgpr_spill_count"] == 0 and r["private_segment_fixed_size"] <= (128 if C == 8 else 64), (M, C, r)
            # round 4 (SoA accesses as buffer accesses: no 64-bit address pairs): the laws without Euler-angle feedback -- none,
            # position, fully actuated, external wrench -- fit 4 waves per SIMD, the acceleration law takes 2, the rest 3
            assert r["vgpr_count"] <= (128 if C in (0, 1, 7, 8) else (256 if C == 5 else 168)), (M, C, r["vgpr_count"])
        elif not single:
            assert r["vgpr_count"] <= 256  # 2 waves per SIMD
            assert r["vgpr_spill_count"] <= 40, (M, C, r)  # M = 8 only; n > 65536 envs with k > 1: not a BASELINE config
        if r["vgpr_spill_count"] or r["private_segment_fixed_size"]:
            report.append((M, C, single, wide, r["vgpr_spill_count"], r["private_segment_fixed_size"]))
    print("k_env_step instances with spills or a private segment (M, ctrl, single, wide, vgpr spills, bytes):", report)


def test_four_lanes_per_env_kernel(meta):
    """k_env_step_quad_position (BASELINE configs 1/2): 16 envs per wave, compiled for one wave per SIMD; no spill, no scratch,
    no LDS (components travel between the lanes of a quad as DPP operand modifiers)."""
    ks = {n: r for n, r in meta.items() if "k_env_step_quad_position" in n}
    assert len(ks) == 1
    r = next(iter(ks.values()))
    assert r["vgpr_spill_count"] == 0 and r["sgpr_spill_count"] == 0 and r["private_segment_fixed_size"] == 0, r
    assert r["group_segment_fixed_size"] == 0 and r["max_flat_workgroup_size"] == 64 and r["vgpr_count"] <= 128, r
    # the sub-step loop (the six Lee laws of the quadrotor, the fully actuated octarotor) and the reset / observation launch
    loops = {n: r for n, r in meta.items() if "k_env_step_quad_loop<" in n or "k_reset_masked_quad_obs" in n}
    assert len(loops) == 11  # 9 sub-step loops + the reset / observation launch in its device-RNG and host-draw instances
    for name, r in loops.items():
        assert r["vgpr_spill_count"] == 0 and r["private_segment_fixed_size"] == 0 and r["max_flat_workgroup_size"] == 64, (name, r)
        # one-wave workgroups, at most 65 536 envs = 4096 waves: two waves per SIMD keep 32 768 envs resident in one round
        # (round 3: the float64 elementary functions took the Lee-law instances from <= 168 to 161-199 VGPRs)
        assert r["vgpr_count"] <= 256, (name, r["vgpr_count"])


def test_wide_env_step_kernels_issue_no_scratch_instruction():
    """`.private_segment_fixed_size` of some WIDE k-loop instances is non-zero (frame slots the SGPR spiller
    reserved and then did not need); what matters is that no scratch instruction is ever issued."""
    import tempfile

    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        obj = os.path.join(_build.LIB_DIR, "agx_dynamics.o")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
        co = os.path.join(tmp, "dyn.co")
        subprocess.run([os.path.join(codeobj.LLVM_BIN, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                        f"--targets={codeobj.TARGET}", f"--output={co}"], check=True)
        asm = subprocess.run([os.path.join(codeobj.LLVM_BIN, "llvm-objdump"), "-d", co], check=True, capture_output=True,
                             text=True).stdout
    cur, scratch, seen = None, {}, 0
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
            continue
        # (gfx950 spills through scratch_* instructions; buffer_load / buffer_store ... offen are the kernels' own SoA accesses)
        if cur and "k_env_step" in cur and "scratch_" in line:
            scratch[cur] = scratch.get(cur, 0) + 1
        if cur and "k_env_step" in cur and "s_endpgm" in line:
            seen += 1
    assert seen >= 108
    wide = {k: v for k, v in scratch.items() if re.search(r"k_env_stepILi\dELi\dELb[01]ELb1E", k)}
    assert not wide, wide
    single_quad = {k: v for k, v in scratch.items() if re.search(r"k_env_stepILi4ELi\dELb1ELb0E", k)}
    assert not single_quad, single_quad


def test_raycast_kernels_fit_eight_waves_per_simd(meta):
    rays = {n: r for n, r in meta.items() if n.startswith("void agx::k_raycast<")}
    assert len(rays) == 5  # camera {basic, normal, stereo}, LiDAR {basic, normal}; rejected variants live in profiles/src/raycast_variants
    for name, r in rays.items():
        variant = int(re.match(r"void agx::k_raycast<(true|false), (\d)>", name).group(2))
        # (round 5, object nodes: the stereo instance -- two rays' state -- keeps two values in scratch around its traversals, outside
        #  the loops; the camera / LiDAR instances of the benchmark configurations use none)
        assert r["group_segment_fixed_size"] == 0, name
        assert (r["vgpr_spill_count"] <= 4 and r["private_segment_fixed_size"] <= 16) if variant == 2 else \
            (r["vgpr_spill_count"] == 0 and r["private_segment_fixed_size"] == 0), (name, r)
        # BASIC / NORMAL: 8 waves per SIMD; STEREO (two rays' state + the six instances of the triangle test): 6
        assert r["vgpr_count"] <= (80 if variant == 2 else 64), (name, r["vgpr_count"])


def test_no_kernel_of_the_hot_path_uses_scratch(meta):
    """Every other kernel of the library: no scratch at all."""
    # (k_scene_refresh: <= 64 bytes of frame slots the SGPR spiller reserved for its by-value argument structs and did not need --
    #  the disassembly has no scratch_ instruction; vgpr_spill_count == 0 is asserted below)
    bad = {n: r["private_segment_fixed_size"] for n, r in meta.items()
           if r["private_segment_fixed_size"] and "k_env_step" not in n and not n.startswith("void agx::k_raycast<false, 2>")
           and not (n.startswith(("agx::k_scene_refresh", "void agx::k_scene_refresh<")) and r["private_segment_fixed_size"] <= 64
                    and r["vgpr_spill_count"] == 0)}
    # (round 6: each instance of the refresh / build kernels carries ONE tree builder; the object-level instance -- the one the
    #  benchmark configurations launch -- uses no scratch at all)
    obj = [r for n, r in meta.items() if n.startswith(("void agx::k_scene_refresh<true>", "void agx::k_bvh_build<true>"))]
    assert len(obj) == 2 and all(r["private_segment_fixed_size"] == 0 and r["vgpr_spill_count"] == 0 for r in obj), obj
    assert not bad, bad
