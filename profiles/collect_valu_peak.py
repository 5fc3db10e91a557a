"""Turns the output of profiles/src/valu_peak.hip (JSON lines; run plain) and, optionally, the rocprofv3 PMC pass of the same
binary (`--pmc GRBM_GUI_ACTIVE`, counter collection CSV) into profiles/r04_valu_peak.json: the measured VALU issue ceiling and
the effective clock it was reached at (GRBM_GUI_ACTIVE / wall time of the launch, MI355X_MICROARCH.md "DVFS").

    python profiles/collect_valu_peak.py gpurun_out/r03m/valu_peak.jsonl [gpurun_out/r03m/valu_peak_pmc/..._counter_collection.csv] [gpurun_out/r03m/valu_peak_pmc.jsonl]
"""
import csv
import json
import os
import sys


XCDS = 8


def main(plain, pmc_csv=None, pmc_jsonl=None):
    rows = [json.loads(l) for l in open(plain) if l.strip().startswith("{")]
    if pmc_csv and pmc_jsonl and os.path.exists(pmc_csv):
        prof = [json.loads(l) for l in open(pmc_jsonl) if l.strip().startswith("{")]  # wall times OF THE PROFILED run
        gui = {}
        for r in csv.DictReader(open(pmc_csv)):
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and "k_valu" in r["Kernel_Name"]:
                gui.setdefault(int(r["Dispatch_Id"]), 0.0)
                gui[int(r["Dispatch_Id"])] += float(r["Counter_Value"])
        order = sorted(gui)  # dispatch ids of the k_valu launches, in launch order
        for row, p in zip(rows, prof):
            i = p["timed_dispatch_index"]
            if i < len(order):
                cycles = gui[order[i]] / XCDS  # the counter is reported summed over the 8 XCDs (18.2 "GHz" otherwise)
                row["profiled_run"] = {"ms": p["ms"], "GRBM_GUI_ACTIVE_per_xcd": cycles, "effective_clock_ghz": cycles / (p["ms"] * 1e-3) / 1e9,
                                       "wave_instr_per_s": p["wave_instr_per_s"],
                                       "cycles_per_wave_instr_at_effective_clock": cycles * p["cus"] * 4 / p["wave_instr"]}
    best = max(rows, key=lambda r: r["wave_instr_per_s"])
    # round 4: issue cost of the other instruction classes (rows with a "class": 16 independent chains, 8 waves per SIMD), in
    # cycles per wave64 instruction per SIMD at the effective clock of the profiled run when there is one, else at the maximum clock
    def cycles(r):
        return (r.get("profiled_run") or {}).get("cycles_per_wave_instr_at_effective_clock") or r["cycles_per_wave_instr_at_max_clock"]
    class_cycles, class_rate = {}, {}
    for r in rows:
        if "class" in r:
            class_cycles.setdefault(r["class"], {})[r["kind"]] = cycles(r)
            class_rate.setdefault(r["class"], {})[r["kind"]] = r["wave_instr_per_s"]
    plain = cycles(best)
    out = {"wave_instr_per_s": best["wave_instr_per_s"], "best": best, "plain_cycles_per_wave_instr": plain,
           "class_cycles_per_wave_instr": class_cycles, "class_wave_instr_per_s": class_rate,
           "effective_clock_ghz": (best.get("profiled_run") or {}).get("effective_clock_ghz"),
           "guide": {"cycles_per_wave_instr": 2.0, "max_clock_ghz": 2.4, "wave_instr_per_s": 256 * 4 * 2.4e9 / 2.0,
                     "source": "MI355X_MICROARCH.md, per-instruction cycle constants: v_fma_f32 (wave64) 2 cyc (SIMD-32)"},
           "rows": rows, "source": "profiles/src/valu_peak.hip on one MI355X"}
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("AGX_VALU_PEAK_OUT", "r04_valu_peak.json"))
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "rows"}, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
