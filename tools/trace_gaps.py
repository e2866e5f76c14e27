"""Per-step timeline from a rocprofv3 kernel trace of bench.py: for every decode kernel, the gap
before it and what ran between two decode kernels.   python tools/trace_gaps.py kernel_trace.csv"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')))
rows.sort()
dec = [i for i, r in enumerate(rows) if 'paged_decode_kernel' in r[2]]
per = []
for a, b in zip(dec[-60:-1], dec[-59:]):
    per.append((rows[b][0] - rows[a][0]) / 1e3)
print(f"decode start-to-start over the last {len(per)} steps: mean {sum(per) / len(per):.1f} us")
a, b = dec[-3], dec[-2]
t0 = rows[a][0]
for r in rows[a:b + 1]:
    print(f"  +{(r[0] - t0) / 1e3:8.1f} us  dur {(r[1] - r[0]) / 1e3:7.1f} us  q{r[3]}  {r[2][:90]}")
