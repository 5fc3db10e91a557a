"""Turns the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as the MI355X guide
prescribes) of profiles/pmc_probe.py into per-launch HBM traffic, calibrated on a kernel with a
known byte count in the same access pattern, and writes profiles/pmc_traffic.json.

    python profiles/collect_pmc.py gpurun_out/pmc_fetch/p_counter_collection.csv gpurun_out/pmc_write/p_counter_collection.csv
"""
import csv
import json
import os
import sys
from collections import defaultdict


def load(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        key = None
        if "k_update_states" in name:
            key = ("k_update_states", int(r["Grid_Size"]))
        elif "k_env_step" in name:
            import re

            m = re.search(r"(k_env_step\w*)(<[^>]*>)?", name)  # k_env_step<motors, ctrl, single, wide> | k_env_step_quad_*
            key = (m.group(1) + (m.group(2) or "").replace(" ", ""), int(r["Grid_Size"]))
        elif "k_raycast" in name:
            import re

            m = re.search(r"k_raycast<([^>]*)>", name)  # <lidar, variant>
            key = ("k_raycast<%s>" % m.group(1).replace(" ", "") if m else "k_raycast", int(r["Grid_Size"]))
        elif "k_reset_masked" in name:
            import re

            key = (re.search(r"k_reset_masked\w*", name).group(0), int(r["Grid_Size"]))  # k_reset_masked<..> | k_reset_masked_quad_obs
        if key:
            acc[key].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


SQ_COUNTERS = ("SQ_INSTS_VALU", "SQ_WAVES", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY",
               "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
               "SQ_INSTS_LDS", "SQ_ACTIVE_INST_SCA", "SQ_INST_CYCLES_SALU", "SQ_INSTS_SALU")
# round 4: the instruction CLASSES of the vector instructions (two more passes): what the vector pipe is busy with is not a count
# of instructions against the plain-fp32 issue ceiling when a kernel evaluates functions in float64, divides and runs Philox
VALU_CLASSES = ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32",
                "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64",
                "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT")


def main(fetch_csv, write_csv, sq_csv=None, build_id=None, sq2_csv=None, sq3_csv=None, sq4_csv=None):
    """build_id: what the library the counters were collected on reported (agx_build_id; pmc_probe.py writes it next to
    the CSVs): bench.py marks every number taken from this file stale when the library it runs on says something else."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from aerial_gym_simulator_amd import _build

    fetch, nf = load(fetch_csv, "FETCH_SIZE")
    write, nw = load(write_csv, "WRITE_SIZE")
    KB = 1024.0
    if build_id and os.path.exists(build_id):
        build_id = open(build_id).read().strip()
    out = {"unit": "bytes per launch", "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), counters in KiB",
           "source_hash": _build.source_hash(), "build_id": build_id or _build.source_hash(),
           "rule": "traffic = FETCH_SIZE x fetch_correction + WRITE_SIZE x write_correction for EVERY kernel of this file, the two "
                   "corrections measured in the same passes on a kernel of known traffic (k_update_states at 2^21 envs: 52 B read, "
                   "64 B written per env, 4 B per lane coalesced); `<kernel>_detail.guide_rule` = 2 x FETCH_SIZE + WRITE_SIZE, the "
                   "reading MI355X_MICROARCH.md gives for wide (16 B per lane) streaming reads, for comparison"}
    if sq_csv:  # third pass: vector instructions issued per launch (wave64 instructions), for the VALU-issue roofline
        valu, nv = load(sq_csv, "SQ_INSTS_VALU")
        waves, _ = load(sq_csv, "SQ_WAVES")
        out["valu_wave_instructions"] = {"%s_%d" % k: v for k, v in valu.items()}
        out["waves"] = {"%s_%d" % k: v for k, v in waves.items()}
        out["valu_source"] = "rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES ... (own pass), average over %s launches" % sorted(set(nv.values()))
        # where the issue slots go: every SQ counter of the pass(es), per launch, for the ray-cast and
        # env-step kernels; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves
        sq = {}
        for path in (sq_csv, sq2_csv, sq3_csv, sq4_csv):
            if not path or not os.path.exists(path):
                continue
            for name in SQ_COUNTERS + VALU_CLASSES:
                vals, _n = load(path, name)
                for k, v in vals.items():
                    if k[0].startswith(("k_raycast", "k_env_step")):
                        sq.setdefault("%s_%d" % k, {})[name] = v
        for k, d in sq.items():
            wc = d.get("SQ_WAVE_CYCLES")
            if wc:
                for name in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_SCA", "SQ_INST_CYCLES_SALU"):
                    if name in d:
                        d[name + "_over_WAVE_CYCLES"] = d[name] / wc
            if d.get("SQ_INSTS_VALU"):
                for name in ("SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS"):
                    if name in d:
                        d[name + "_per_VALU"] = d[name] / d["SQ_INSTS_VALU"]
                if d.get("SQ_ACTIVE_INST_VALU"):
                    d["quad_cycles_per_VALU_instruction"] = d["SQ_ACTIVE_INST_VALU"] / d["SQ_INSTS_VALU"]
                mix = {c[len("SQ_INSTS_VALU_"):]: d[c] for c in VALU_CLASSES if c in d}
                if mix:
                    mix["other (moves, selects, compares, bit operations, DPP: not in any class counter)"] = d["SQ_INSTS_VALU"] - sum(mix.values())
                    d["valu_class_mix"] = mix
                    d["valu_class_share"] = {k: v / d["SQ_INSTS_VALU"] for k, v in mix.items()}
        out["sq_breakdown"] = sq
    n_cal = 1 << 21
    cal = ("k_update_states", n_cal)
    if cal in fetch and cal in write:
        # known traffic of k_update_states: reads 13 floats, writes 16 floats per env
        rd_true, wr_true = 13 * 4 * n_cal, 16 * 4 * n_cal
        cf, cw = rd_true / (fetch[cal] * KB), wr_true / (write[cal] * KB)
        out["calibration"] = {"kernel": "k_update_states @ 2^21 envs (4 B/lane coalesced SoA)", "fetch_raw_bytes": fetch[cal] * KB,
                              "fetch_true_bytes": rd_true, "fetch_correction": cf, "write_raw_bytes": write[cal] * KB,
                              "write_true_bytes": wr_true, "write_correction": cw}
    else:
        cf = cw = 1.0
    for (name, grid) in sorted(k for k in fetch if k[0].startswith(("k_env_step", "k_reset_masked"))):
        tag = "%s_%d" % (name, grid)
        if (name, grid) in fetch and (name, grid) in write:
            raw = fetch[(name, grid)] * KB + write[(name, grid)] * KB
            out[tag] = fetch[(name, grid)] * KB * cf + write[(name, grid)] * KB * cw
            out[tag + "_detail"] = {"fetch_raw": fetch[(name, grid)] * KB, "write_raw": write[(name, grid)] * KB, "raw_sum": raw,
                                    "launches_averaged": [nf[(name, grid)], nw[(name, grid)]]}
    for (name, grid), v in fetch.items():
        if name.startswith("k_raycast") and (name, grid) in write:
            tag = "%s_%d" % (name, grid)  # grid size in threads = envs x sensors x 256 (x tile split)
            # the SAME rule as for the env-step kernels (one calibration per file); node / triangle reads are scalar
            # (wave-uniform) loads, for which neither this calibration nor the guide's x2 is established: both are listed
            out[tag] = v * KB * cf + write[(name, grid)] * KB * cw
            out[tag + "_detail"] = {"fetch_raw": v * KB, "write_raw": write[(name, grid)] * KB,
                                    "guide_rule": 2.0 * v * KB + write[(name, grid)] * KB}
    for tag in [k for k in out if k.endswith("_detail") and "guide_rule" not in out[k]]:
        out[tag]["guide_rule"] = 2.0 * out[tag]["fetch_raw"] + out[tag]["write_raw"]
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_traffic.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:8])
