"""Gaps between consecutive kernels from a rocprofv3 --kernel-trace csv.  usage: python tools/chain_timeline.py <dir> [first_row] [rows]"""
import csv, glob, os, sys
d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
n0 = int(sys.argv[2]) if len(sys.argv) > 2 else max(0, len(rows) - 60)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
prev_end = None
for r in rows[n0:n0 + n]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").replace("flimo::", "")[:48]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%-48s gap %7.2f us  dur %7.2f us" % (name, gap, (e - s) / 1e3))
    prev_end = e
