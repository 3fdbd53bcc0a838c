"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per dispatch of each kernel."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if "rocclr" in k:
                continue
            print(k, {c: (sum(v) / len(v)) for c, v in cs.items()}, "dispatches", len(next(iter(cs.values()))))
