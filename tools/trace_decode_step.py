"""Kernels of one captured decode step of the engine, from a rocprofv3 kernel trace of
tools/bench_engine.py: everything between two consecutive decode_inputs_kernel launches near the end
in the middle
of the run's decode phase, with durations and gaps.   python tools/trace_decode_step.py kernel_trace.csv"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
marks = [i for i, r in enumerate(rows) if 'decode_inputs_kernel' in r[2]]
# the pure-decode step with the most attention time (the run's last full-batch steps, replayed
# from the captured graph); steps that hold prompt chunks and the captures' eager warm-ups are skipped
best, a, b = -1, marks[0], marks[1]
for i, j in zip(marks[:-1], marks[1:]):
    names = [r[2] for r in rows[i:j]]
    if any('prefill' in n for n in names) or (rows[j][0] - rows[i][0]) > 1.2e6:
        continue
    att = sum(r[1] - r[0] for r in rows[i:j] if 'paged_decode_kernel' in r[2])
    if att > best:
        best, a, b = att, i, j
step = rows[a:b]
print(f"{len(step)} kernels, {(rows[b][0] - rows[a][0]) / 1e3:.1f} us start to start, "
      f"{sum(r[1] - r[0] for r in step) / 1e3:.1f} us inside kernels")
prev_end = None
for s, e, name in step:
    short = name.replace('void ', '').replace('mi355::(anonymous namespace)::', '').replace('(anonymous namespace)::', '')
    short = short.split('(')[0].split('<')[0][:60]
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"  gap {gap:5.1f}  dur {(e - s) / 1e3:6.1f}  {short}")
    prev_end = e
