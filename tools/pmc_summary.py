"""Summarise rocprofv3 --pmc passes (separate FETCH_SIZE and WRITE_SIZE runs of the same bench command) into
profiles/<round>/pmc_fetch_write_per_kernel.json.  Units and gfx950 corrections as MI355X_MICROARCH.md prescribes:
FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE on gfx950 reports 1/2 of a wide coalesced read -> doubled;
WRITE_SIZE is taken as reported (uncalibrated).
usage: python tools/pmc_summary.py <dir with fetch pass> <dir with write pass> <out.json>"""
import csv, glob, json, os, sys
from collections import defaultdict


def per_kernel(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        # rocprofv3 emits one row per (dispatch, counter [, dimension instance]); sum the instances of a dispatch
        per_dispatch = defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            key = r.get("Dispatch_Id") or r.get("Correlation_Id")
            per_dispatch[key] += float(r["Counter_Value"])
            names[key] = r["Kernel_Name"]
        for k, v in per_dispatch.items():
            acc[names[k]].append(v)
    return acc


def short(name):
    n = name.replace("void ", "")
    return n.split("(")[0]


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    res = {}
    for label, d, counter in (("FETCH_SIZE_KB_per_launch", fetch_dir, "FETCH_SIZE"), ("WRITE_SIZE_KB_per_launch", write_dir, "WRITE_SIZE")):
        acc = per_kernel(d, counter)
        res[label] = {short(k): {"launches": len(v), "mean": sum(v) / len(v)} for k, v in acc.items() if "rocprim" not in k and "hipcub" not in k}
    def find(tbl, key):
        # a pass of a chained update is the same kernel body under another name (knn5_chain_kernel / fit2_chain_kernel): the two are
        # pooled, weighted by their launches
        hits = [v for k, v in tbl.items() if key in k or key.replace("_kernel<", "_chain_kernel<") in k]
        if not hits:
            return None
        n = sum(h["launches"] for h in hits)
        return {"launches": n, "mean": sum(h["mean"] * h["launches"] for h in hits) / n}
    for kern, label in (("knn5_kernel<2, 8, true, false>", "dominant_kernel"), ("knn5_kernel<2, 8, false, false>", "knn5_separate"),
                        ("widen_kernel", "widen_kernel"), ("fit2_kernel", "fit2_kernel"), ("fit_kernel", "fit_kernel")):
        f = find(res["FETCH_SIZE_KB_per_launch"], kern)
        w = find(res["WRITE_SIZE_KB_per_launch"], kern)
        if f and w:
            res[label + "_traffic_bytes_per_launch"] = {
                "kernel": kern, "fetch_raw": f["mean"] * 1024.0, "fetch_corrected_x2": 2.0 * f["mean"] * 1024.0, "write_raw": w["mean"] * 1024.0,
                "total_corrected": 2.0 * f["mean"] * 1024.0 + w["mean"] * 1024.0, "launches": f["launches"],
                "note": "FETCH_SIZE doubled (gfx950 rocprofv3 reports half of wide coalesced reads); WRITE_SIZE as reported"}
    # the profile is only valid for the sources it was taken with (bench.py checks this hash)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from fast_limo_amd import build as b
    res["sources_hash"] = b.sources_hash()
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res.items():
        if k.endswith("_traffic_bytes_per_launch"):
            print(k, json.dumps(v))


if __name__ == "__main__":
    main()
