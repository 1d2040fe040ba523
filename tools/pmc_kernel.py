"""tools/pmc_kernel.py <kernel substring> <counter_collection.csv>...: sums every counter of rocprofv3 --pmc passes over the
dispatches of one kernel (per dispatch averages), to see what a kernel other than k_sweep is bound by.  SQ cycle counters are
in quad-cycles summed over the 8 XCDs; GRBM_GUI_ACTIVE / 8 is the launch's GPU cycles."""
import csv
import json
import sys


def main():
    kernel = sys.argv[1]
    out = {}
    for path in sys.argv[2:]:
        disp = set()
        acc = {}
        for r in csv.DictReader(open(path)):
            if kernel not in r["Kernel_Name"]:
                continue
            disp.add(r["Dispatch_Id"])
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        n = max(len(disp), 1)
        for k, v in acc.items():
            out[k] = v / n
        out.setdefault("dispatches", n)
    print(json.dumps({"kernel": kernel, "per_dispatch": out}, indent=1))


if __name__ == "__main__":
    main()
