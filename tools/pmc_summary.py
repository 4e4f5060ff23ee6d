"""Averages rocprofv3 counter_collection CSVs (tools/pmc_sets.sh) per kernel and counter -> <dir>/summary.json + a table."""
import collections, csv, glob, json, os, sys
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "p*_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items() if k.startswith("kws") or "kws_" in k}
json.dump(res, open(os.path.join(d, "summary.json"), "w"), indent=1)
for k, cs in res.items():
    print("==", k)
    for c, v in sorted(cs.items()):
        print("   %-32s %16.1f" % (c, v))
