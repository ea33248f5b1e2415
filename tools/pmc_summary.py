"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files.  Usage: pmc_summary.py <dir> [<dir> ...]"""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for fn in glob.glob(d + "/*/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(fn)):
            name = row["Kernel_Name"].split("(")[0][-40:]
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for name, cs in acc.items():
            if not ("k2_" in name or "k3_" in name or "k_pass" in name or "k_octant" in name or "k_map" in name):
                continue
            print(name, {c: f"{sum(v)/len(v):.4g}" for c, v in cs.items()})
