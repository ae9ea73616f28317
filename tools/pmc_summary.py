"""Summarise a rocprofv3 --pmc counter_collection csv: per kernel name, mean of each counter per dispatch."""
import csv, sys, collections, glob
path = sys.argv[1]
files = glob.glob(path + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        short = name.split("::")[1].split("(")[0] if "anonymous" in name else name[:40]
        if len(sys.argv) > 2 and sys.argv[2] not in short:
            continue
        agg[(short, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k, {c: (sum(v) / len(v), len(v)) for c, v in d.items()})
