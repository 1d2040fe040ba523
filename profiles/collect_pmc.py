"""Turns the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes)
into profiles/pmc_traffic.json, which bench.py reports as roofline.traffic.

    python profiles/collect_pmc.py gpurun_out/pmc_r01/fetch_counter_collection.csv gpurun_out/pmc_r01/write_counter_collection.csv

Per k_sweep launch of the TIMED steps (the last 2*steps launches): hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide streaming reads, hence the
factor 2 (MI355X_MICROARCH.md, HBM section).  The sweep's reads are 8-byte gathers, a width the guide marks as
uncalibrated, so the read side is an upper-bound style estimate; both raw counters are kept in the file."""
import csv
import json
import os
import sys


def per_launch(path, counter, kernel="k_sweep", last=6):
    vals = {}
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
            vals[int(r["Dispatch_Id"])] = vals.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    v = [vals[k] for k in sorted(vals)][-last:]
    return v


def main():
    fetch_csv, write_csv = sys.argv[1], sys.argv[2]
    last = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    f = per_launch(fetch_csv, "FETCH_SIZE", last=last)
    w = per_launch(write_csv, "WRITE_SIZE", last=last)
    fk, wk = sum(f) / len(f), sum(w) / len(w)
    out = {"kernel": "k_sweep", "launches": len(f), "FETCH_SIZE_KiB_per_launch": fk, "WRITE_SIZE_KiB_per_launch": wk,
           "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024.0,
           "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 3 --warmup 1`; "
                   "FETCH_SIZE doubled per the gfx950 correction; 8-byte gathers are an uncalibrated width"}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_traffic.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
