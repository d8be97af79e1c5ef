"""Summarise a rocprofv3 --pmc run: per kernel name, mean duration, effective clock (GRBM_GUI_ACTIVE / duration) and
MFMA-pipe busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 4 SIMDs * 256 CUs))."""
import csv, sys, glob, collections
d = sys.argv[1]
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        cnt[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        cnt[k]["_dur"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k, c in sorted(cnt.items(), key=lambda kv: -sum(kv[1]["_dur"])):
    dur = sum(c["_dur"]) / len(c["_dur"])
    line = "%-60s n=%4d dur %9.1f us" % (k, len(c["_dur"]), dur / 1e3)
    if "GRBM_GUI_ACTIVE" in c:
        g = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"])
        line += "  clk %.2f GHz" % (g / dur)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            m = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
            line += "  mfma_busy/(gui*1024) %.3f" % (m / (g * 1024))
    for name in c:
        if name not in ("_dur", "GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES"):
            line += "  %s=%.3g" % (name, sum(c[name]) / len(c[name]))
    print(line)
