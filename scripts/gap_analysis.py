"""Idle time between consecutive kernels of the last 10-iteration run in a rocprofv3 --kernel-trace CSV.
   python scripts/gap_analysis.py <..._kernel_trace.csv> [span_ms]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
span = float(sys.argv[2]) if len(sys.argv) > 2 else 12.4
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("cubahip::", "").replace("void ", "")) for r in rows))
tend = ev[-1][1]; t0 = tend - int(span * 1e6)
run = [e for e in ev if e[0] >= t0]
busy = sum(e[1] - e[0] for e in run)
print("events", len(run), "span ms", (run[-1][1] - run[0][0]) / 1e6, "busy ms", busy / 1e6)
gaps, gapn = collections.Counter(), collections.Counter()
for a, b in zip(run, run[1:]):
    key = a[2][:24] + " -> " + b[2][:24]
    gaps[key] += b[0] - a[1]; gapn[key] += 1
print("total gap ms", sum(gaps.values()) / 1e6)
for k, v in gaps.most_common(16):
    print(f"{v/1e3:9.1f} us  n={gapn[k]:4d}  avg {v/gapn[k]/1e3:6.2f}  {k}")
dur, durn = collections.Counter(), collections.Counter()
for e in run: dur[e[2]] += e[1] - e[0]; durn[e[2]] += 1
print("kernel time:")
for k, v in dur.most_common(14): print(f"{v/1e3:9.1f} us  n={durn[k]:4d}  avg {v/durn[k]/1e3:7.2f}  {k}")
