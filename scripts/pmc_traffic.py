"""Per-kernel HBM-side traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB per launch as reported).
   python scripts/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass>  > profiles/..._pmc_traffic.json
FETCH_SIZE under-counts wide coalesced streams by 2x on gfx950 (MI355X_MICROARCH.md): both the raw sum and the
fetch-doubled sum are given."""
import csv, glob, json, os, re, sys
from collections import defaultdict


def collect(d, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("cubahip::", "").replace("void ", "").strip()
            tot[name] += float(row["Counter_Value"]); cnt[name] += 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


fetch = collect(sys.argv[1], "FETCH_SIZE")
write = collect(sys.argv[2], "WRITE_SIZE")
def source_sha16():
    """hash of the device-side sources this profile was taken with: every .hip / .hpp under csrc/ (same function as bench.py's;
    bench.py flags a traffic file from other sources as stale)"""
    import hashlib
    h = hashlib.sha256()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "cuda-bundle-adjustment_amd", "csrc")
    for f in sorted(f for f in os.listdir(csrc) if f.endswith((".hip", ".hpp"))):
        with open(os.path.join(csrc, f), "rb") as fh:
            h.update(f.encode()); h.update(fh.read())
    return h.hexdigest()[:16]


out = {"kernel_source_sha16": source_sha16(), "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes), python scripts/prof_run.py kitti00 1; "
               "KiB per launch as reported; FETCH_SIZE under-counts wide coalesced streams by 2x on gfx950 (MI355X_MICROARCH.md), "
               "other access widths uncalibrated -> both the raw sum and the fetch-doubled sum are given",
       "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    f, n = fetch.get(k, (0.0, 0)); w, n2 = write.get(k, (0.0, 0))
    out["kernels"][k] = {"launches": max(n, n2), "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w,
                         "hbm_bytes_raw": 1024 * (f + w), "hbm_bytes_fetch_x2": 1024 * (2 * f + w)}
print(json.dumps(out, indent=1))
