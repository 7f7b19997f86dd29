#!/usr/bin/env python3
"""Developer tool: GPU timeline of ONE sweep of the end-to-end leg (front end, passes, map insert) from a rocprofv3 kernel trace.

  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline \
      --no-hbm-regime --streams 0
  python3 tools/sweep_timeline.py gpurun_out/tl

Prints every kernel between the last two launches of the anchor kernel (default: bbox_finite_kernel = start of a map insert):
start offset, duration, idle gap before it."""
import csv, glob, sys, os

def main():
    d = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 else "bbox_finite_kernel"
    back = int(sys.argv[3]) if len(sys.argv) > 3 else 1          # which cycle from the end (1 = last complete one)
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    idx = [i for i, r in enumerate(rows) if anchor in r[2]]
    if len(idx) < back + 1:
        print("anchor seen", len(idx), "times"); return
    a, b = idx[-back - 1], idx[-back]
    t0 = rows[a][0]
    prev_end = rows[a][0]
    busy = 0
    for s, e, name in rows[a:b]:
        short = name.split("(")[0].replace("void ", "").replace("flimo::", "")
        if "rocprim" in short:
            short = "rocprim:" + short.split("::")[-1][:40]
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:7.1f}  {short[:70]}")
        prev_end = max(prev_end, e)
        busy += e - s
    print(f"cycle {(rows[b][0] - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, {b - a} launches")

if __name__ == "__main__":
    main()
