import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = lambda r: r["Kernel_Name"][:40]
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for a, b in zip(rows[:-1], rows[1:]):
    gap[names(b)].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
for r in rows: dur[names(r)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
import statistics as st
for k in dur:
    if len(dur[k]) > 50:
        print(f"{k:42s} n={len(dur[k]):5d} dur median {st.median(dur[k])/1e3:8.2f} us  gap-before median {st.median(gap[k])/1e3 if gap[k] else 0:8.2f} us")
