"""tools/pmc_sq.py <counter_collection.csv> [kernel]: per-launch SQ summary of a rocprofv3 --pmc pass with
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE
(SQ cycle counters are in quad-cycles, summed over the 8 XCDs; 1024 SIMDs)."""
import csv
import json
import sys


def main():
    path = sys.argv[1]
    kernel = sys.argv[2] if len(sys.argv) > 2 else "k_sweep"
    d = {}
    for r in csv.DictReader(open(path)):
        if kernel not in r["Kernel_Name"]:
            continue
        k = int(r["Dispatch_Id"])
        d.setdefault(k, {})
        d[k][r["Counter_Name"]] = d[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    out = []
    for k in sorted(d):
        c = d[k]
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0  # per XCD
        if gui <= 0:
            continue
        simd_cycles = gui * 1024.0
        row = {"dispatch": k,
               "gpu_cycles": gui,
               "waves_per_SIMD": 4.0 * c.get("SQ_WAVE_CYCLES", 0.0) / simd_cycles,
               "valu_busy_frac_of_SIMD_time": 4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / simd_cycles,
               "wait_any_frac_of_wave_time": c.get("SQ_WAIT_ANY", 0.0) / max(c.get("SQ_WAVE_CYCLES", 1.0), 1.0),
               "wait_inst_frac": c.get("SQ_WAIT_INST_ANY", 0.0) / max(c.get("SQ_WAVE_CYCLES", 1.0), 1.0),
               "valu_insts": c.get("SQ_INSTS_VALU", 0.0)}
        out.append(row)
    print(json.dumps({"kernel": kernel, "launches": out}, indent=1))


if __name__ == "__main__":
    main()
