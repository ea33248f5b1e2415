"""HBM bytes per launch of the transform pass kernels (k_passB includes the in-place middle-axis passes of the sandwich) from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE)
of the SAME bench.py command.  FETCH_SIZE is doubled (gfx950 correction of MI355X_MICROARCH.md); both counters are KiB.
Usage: pmc_traffic.py <fetch_dir> <write_dir> <workload> > profiles/<name>_pmc_traffic.json"""
import csv, glob, json, os, sys, collections

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def family(name):
    if "k2_final2" in name or "k_passC2" in name:
        return "k_passC2"
    if "k2_final" in name or "k_passC" in name:
        return "k_passC"
    if "k3_contig" in name:
        return "k_passS1"
    if "k3_mid" in name:
        return "k_passSM"
    if "k2_strided" in name:
        args = name.split("<", 1)[1].split(">")[0].split(",")
        return "k_passA" if args[2].strip() == "3" else "k_passB"
    if "k_passA" in name or "k2_contig" in name:
        return "k_passA"
    if "k_passB" in name:
        return "k_passB"
    return None

def collect(d, counter):
    tot = collections.defaultdict(float); cnt = collections.defaultdict(int); var = collections.defaultdict(lambda: [0, 0.0])
    for fn in glob.glob(d + "/*/*counter_collection.csv"):
        for row in csv.DictReader(open(fn)):
            if row["Counter_Name"] != counter:
                continue
            fam = family(row["Kernel_Name"])
            if fam is None:
                continue
            v = float(row["Counter_Value"])
            tot[fam] += v; cnt[fam] += 1
            key = row["Kernel_Name"].split("(")[0].replace("void ", "")
            var[(fam, key)][0] += 1; var[(fam, key)][1] += v
    return tot, cnt, var

ft, fc, fv = collect(sys.argv[1], "FETCH_SIZE")
wt, wc, wv = collect(sys.argv[2], "WRITE_SIZE")
out = {"note": "HBM bytes per launch (average over all launches of the pass kernel family in one bench step) from separate "
               "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the bench command; FETCH_SIZE doubled per the gfx950 "
               "correction of MI355X_MICROARCH.md; KiB units", "workload": sys.argv[3], "kernels": {}}
try:  # the kernel sources these counters belong to (bench.py quotes them only for the same digest)
    import bench
    out["kernel_source_digest"] = bench.kernel_source_digest()
except Exception as exc:
    out["kernel_source_digest"] = None
    out["digest_error"] = repr(exc)
if len(sys.argv) > 4:
    out["commit"] = sys.argv[4]
for fam in sorted(ft):
    f_kb, w_kb = ft[fam] / fc[fam], wt[fam] / max(1, wc[fam])
    out["kernels"][fam] = {"launches": fc[fam], "fetch_kb": round(f_kb, 1), "write_kb": round(w_kb, 1),
                           "bytes": round((2 * f_kb + w_kb) * 1024),
                           "variants": {k[1]: {"launches": v[0], "fetch_kb": round(v[1] / v[0], 1),
                                               "write_kb": round(wv[k][1] / max(1, wv[k][0]), 1)}
                                        for k, v in sorted(fv.items()) if k[0] == fam}}
print(json.dumps(out, indent=1))
