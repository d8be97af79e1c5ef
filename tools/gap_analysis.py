"""Idle time between consecutive kernels of the train step, from a rocprofv3 --kernel-trace csv (Start_Timestamp / End_Timestamp).
usage: gap_analysis.py <dir with *kernel_trace.csv> [first kernel name fragment of a step]"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
mark = sys.argv[2] if len(sys.argv) > 2 else "input_features"
starts = [i for i, r in enumerate(rows) if mark in r[2]]
print("kernels", len(rows), "steps", len(starts))
for a, b in list(zip(starts, starts[1:]))[-4:]:
    seg = rows[a:b]
    busy = sum(e - s for s, e, _ in seg)
    span = rows[b][0] - seg[0][0]
    gaps = [seg[i + 1][0] - seg[i][1] for i in range(len(seg) - 1)] + [rows[b][0] - seg[-1][1]]
    pos = [g for g in gaps if g > 0]
    # union of the busy intervals (two streams: kernels overlap)
    cur_s, cur_e, union = seg[0][0], seg[0][1], 0
    for s_, e_, _ in seg[1:]:
        if s_ > cur_e: union += cur_e - cur_s; cur_s, cur_e = s_, e_
        else: cur_e = max(cur_e, e_)
    union += cur_e - cur_s
    print("      union of busy intervals %.3f ms, no kernel running %.3f ms" % (union / 1e6, (span - union) / 1e6))
    print("step: %d kernels, span %.3f ms, busy %.3f ms, idle %.3f ms (%.1f %%), mean gap %.2f us, max gap %.1f us, overlapped %d" % (
        len(seg), span / 1e6, busy / 1e6, sum(pos) / 1e6, 100.0 * sum(pos) / span, sum(pos) / max(len(pos), 1) / 1e3, max(gaps) / 1e3,
        sum(1 for g in gaps if g < 0)))
