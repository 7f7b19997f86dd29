"""Developer script: per-dispatch durations of one kernel, in launch order, from a rocprofv3 --kernel-trace CSV.
usage: python tools/trace_seq.py <dir with *_kernel_trace.csv> <kernel name prefix> [max rows]"""
import csv, glob, os, sys
d, pref = sys.argv[1], sys.argv[2]
lim = int(sys.argv[3]) if len(sys.argv) > 3 else 80
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].replace("void ", "").replace("flimo::", "").startswith(pref)]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
out = ["%.0f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows[:lim]]
print(pref, "us:", " ".join(out))
