"""tools/collect_profiles.py [round]: turns gpurun_out/profiles_new/ (written by tools/refresh_profiles.sh on a GPU box) into the
files profiles/ keeps: bench lines, kernel stats, the k_sweep rows of the kernel trace, pmc_traffic.json (what bench.py quotes as
roofline.traffic), the per-kernel traffic table of the --filter run and the SQ summary."""
import collections
import csv
import io
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = os.path.join(ROOT, "gpurun_out", "profiles_new")
P = os.path.join(ROOT, "profiles")
R = sys.argv[1] if len(sys.argv) > 1 else "r02"


def per_launch(path, counter, kernel, last):
    vals = {}
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
            vals[int(r["Dispatch_Id"])] = vals.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    return [vals[k] for k in sorted(vals)][-last:]


def agg(path, counter):
    d = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[r["Kernel_Name"].split("(")[0]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    return d


def main():
    for src, dst in ((f"{R}_bench_1gpu.json", f"{R}_bench_1gpu.json"), (f"{R}_bench_filter.json", f"{R}_bench_filter.json"),
                     ("kt/kt_kernel_stats.csv", f"{R}_kernel_stats.csv"), ("ktf/ktf_kernel_stats.csv", f"{R}_kernel_stats_filter.csv"),
                     (f"{R}_pmc_sq_k_sweep.json", f"{R}_pmc_sq_k_sweep.json")):
        shutil.copy(os.path.join(N, src), os.path.join(P, dst))
    rows = [r for r in csv.DictReader(open(os.path.join(N, "kt/kt_kernel_trace.csv"))) if "k_sweep" in r["Kernel_Name"]]
    out = io.StringIO()
    w = csv.DictWriter(out, fieldnames=list(rows[0].keys()))
    w.writeheader()
    for r in rows:
        w.writerow(r)
    open(os.path.join(P, f"{R}_kernel_trace_k_sweep.csv"), "w").write(out.getvalue())
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    print("k_sweep launches ms", [round(x, 1) for x in d], "timed avg", sum(d[-6:]) / 6)
    f = per_launch(os.path.join(N, "fetch/fetch_counter_collection.csv"), "FETCH_SIZE", "k_sweep", 6)
    wr = per_launch(os.path.join(N, "write/write_counter_collection.csv"), "WRITE_SIZE", "k_sweep", 6)
    fk, wk = sum(f) / len(f), sum(wr) / len(wr)
    b = json.load(open(os.path.join(N, f"{R}_bench_1gpu.json")))
    res = {"kernel": "k_sweep", "launches": len(f), "FETCH_SIZE_KiB_per_launch": fk, "WRITE_SIZE_KiB_per_launch": wk,
           "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024.0, "algorithmic_bytes_per_launch": b["roofline"]["algorithmic_bytes_per_launch"],
           "command": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over `bench.py --steps 3 --warmup 1 --cpu-seconds 0` ({R} build)",
           "note": "the six k_sweep launches of the timed 3-iteration schedule; FETCH_SIZE doubled per the gfx950 correction (MI355X_MICROARCH.md, "
                   "HBM section); 8-byte gathers are an uncalibrated width"}
    json.dump(res, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))
    ff, ww = agg(os.path.join(N, "fetchf/fetchf_counter_collection.csv"), "FETCH_SIZE"), agg(os.path.join(N, "writef/writef_counter_collection.csv"), "WRITE_SIZE")
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(N, "ktf/ktf_kernel_trace.csv"))):
        dur[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    lines = ["kernel,calls,ms_per_call,fetch_GB_x2_per_call,write_GB_per_call,GBps"]
    for k in sorted(dur, key=lambda k: -sum(dur[k]))[:16]:
        n = len(dur[k])
        ms = sum(dur[k]) / n
        fg = 2 * sum(ff[k].values()) / max(len(ff[k]), 1) * 1024 / 1e9
        wg = sum(ww[k].values()) / max(len(ww[k]), 1) * 1024 / 1e9
        lines.append('"%s",%d,%.3f,%.2f,%.2f,%.0f' % (k, n, ms, fg, wg, (fg + wg) / ms * 1e3 if ms > 0 else 0))
    open(os.path.join(P, f"{R}_pmc_traffic_per_kernel_filter_run.csv"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:8]))
    print("bench", b["value"], b["roofline"]["frac"], b["ms_per_step"], b["cpu_baseline"]["value"], b["cpu_baseline"]["all_cores"]["value"])
    bf = json.load(open(os.path.join(N, f"{R}_bench_filter.json")))
    print("filter", bf["value"], json.dumps(bf["roofline_filter"]["stage_ms"]), bf["roofline_filter"]["filterExact"]["frac"], bf["roofline_filter"]["filterNeighbor"]["frac"])


if __name__ == "__main__":
    main()
