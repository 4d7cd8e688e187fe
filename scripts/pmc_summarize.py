"""Summarise rocprofv3 --pmc CSVs: mean counter value per dispatch, per kernel."""
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if "nvca" not in k:
        continue
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-36s mean %.6g  (n=%d)" % (c, sum(v) / len(v), len(v)))
