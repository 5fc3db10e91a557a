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

            m = re.search(r"k_raycast<([^>]*)>", name)  # <lidar, lds, variant>
            key = ("k_raycast<%s>" % m.group(1).replace(" ", "") if m else "k_raycast", int(r["Grid_Size"]))
        elif "k_reset_masked" in name:
            key = ("k_reset_masked", int(r["Grid_Size"]))
        if key:
            acc[key].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main(fetch_csv, write_csv, sq_csv=None):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from aerial_gym_simulator_amd import _build

    fetch, nf = load(fetch_csv, "FETCH_SIZE")
    write, nw = load(write_csv, "WRITE_SIZE")
    KB = 1024.0
    out = {"unit": "bytes per launch", "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), counters in KiB",
           "source_hash": _build.source_hash()}
    if sq_csv:  # third pass: vector instructions issued per launch (wave64 instructions), for the VALU-issue roofline
        valu, nv = load(sq_csv, "SQ_INSTS_VALU")
        waves, _ = load(sq_csv, "SQ_WAVES")
        out["valu_wave_instructions"] = {"%s_%d" % k: v for k, v in valu.items()}
        out["waves"] = {"%s_%d" % k: v for k, v in waves.items()}
        out["valu_source"] = "rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES (own pass), average over %s launches" % sorted(set(nv.values()))
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
    for (name, grid) in sorted(k for k in fetch if k[0].startswith("k_env_step")):
        tag = "%s_%d" % (name, grid)
        if (name, grid) in fetch and (name, grid) in write:
            raw = fetch[(name, grid)] * KB + write[(name, grid)] * KB
            out[tag] = fetch[(name, grid)] * KB * cf + write[(name, grid)] * KB * cw
            out[tag + "_detail"] = {"fetch_raw": fetch[(name, grid)] * KB, "write_raw": write[(name, grid)] * KB, "raw_sum": raw,
                                    "launches_averaged": [nf[(name, grid)], nw[(name, grid)]]}
    for (name, grid), v in fetch.items():
        if name.startswith("k_raycast") and (name, grid) in write:
            tag = "%s_%d" % (name, grid)  # grid size in threads = envs x sensors x 256 (x tile split)
            # node / triangle reads are scalar (wave-uniform) loads: the coalesced-dword calibration does not
            # apply to them; report the raw counters and the FETCH x2 reading of the MI355X guide
            out[tag] = 2.0 * v * KB + write[(name, grid)] * KB
            out[tag + "_detail"] = {"fetch_raw": v * KB, "write_raw": write[(name, grid)] * KB,
                                    "note": "traffic = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE under-reports by 2x)"}
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_traffic.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
