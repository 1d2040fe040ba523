"""tools/pmc_mix.py <counter_collection.csv>...: instruction mix of k_sweep from rocprofv3 --pmc passes (sums over the launches of
the run, all waves): prints one JSON object {counter: total} and the shares of SQ_INSTS_VALU."""
import csv
import json
import sys

tot = {}
launches = set()
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        if "k_sweep" not in r["Kernel_Name"]:
            continue
        tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        launches.add(r["Dispatch_Id"])
valu = tot.get("SQ_INSTS_VALU", 0.0)
out = {"kernel": "k_sweep", "launches_summed": len(launches) // max(len(sys.argv) - 1, 1), "totals": tot,
       "share_of_SQ_INSTS_VALU": {k: v / valu for k, v in tot.items() if k.startswith("SQ_INSTS_VALU_") and valu}}
print(json.dumps(out, indent=1))
