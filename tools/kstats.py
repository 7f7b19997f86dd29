"""Print the top kernels of a rocprofv3 --kernel-trace --stats output directory.  usage: python tools/kstats.py <dir> [rows]"""
import csv, glob, os, sys
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True))[0]
for r in list(csv.DictReader(open(f)))[:n]:
    name = r["Name"].replace("void ", "").replace("flimo::", "")[:64]
    print("%-64s calls %5s avg %8.2f us total %9.1f us" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
