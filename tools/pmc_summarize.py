"""Summarise rocprofv3 --pmc CSVs per kernel: mean counter value per dispatch."""
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-40:]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    if "k_trace" not in k and "k_resolve" not in k: continue
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-26s n=%d mean=%.6g" % (c, len(v), sum(v) / len(v)))
