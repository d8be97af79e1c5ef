"""Per-kernel means of every counter in a rocprofv3 --pmc output directory."""
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    print(k, " ".join("%s=%.4g" % (n, sum(v) / len(v)) for n, v in sorted(c.items())), "n=%d" % len(next(iter(c.values()))))
