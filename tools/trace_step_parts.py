"""Where a bench.py step's time sits, from a rocprofv3 kernel trace: average duration of each of the
step's kernels and of the gaps between them, over the steps of the timed region (consecutive
kv-write -> decode -> argmax triples).   python tools/trace_step_parts.py kernel_trace.csv"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
main = [r for r in rows if any(k in r[2] for k in ('kv_write_slots_kernel', 'paged_decode_kernel', 'argmax_kernel'))]
kind = lambda r: 'w' if 'kv_write' in r[2] else 'd' if 'paged_decode' in r[2] else 'a'
steps = []
i = 0
while i + 3 < len(main):
    if [kind(main[i + j]) for j in range(4)] == ['w', 'd', 'a', 'w']:
        steps.append(main[i:i + 4])
        i += 3
    else:
        i += 1
steps = steps[len(steps) // 4: len(steps) - len(steps) // 4]
n = len(steps)
avg = lambda f: sum(f(s) for s in steps) / n / 1e3
print(f"{n} steps: start to start {avg(lambda s: s[3][0] - s[0][0]):7.2f} us")
print(f"  kv write {avg(lambda s: s[0][1] - s[0][0]):6.2f}   gap {avg(lambda s: s[1][0] - s[0][1]):5.2f}   "
      f"decode {avg(lambda s: s[1][1] - s[1][0]):7.2f}   gap {avg(lambda s: s[2][0] - s[1][1]):5.2f}   "
      f"argmax {avg(lambda s: s[2][1] - s[2][0]):6.2f}   gap to the next step {avg(lambda s: s[3][0] - s[2][1]):5.2f}")
others = sorted({r[2][:70] for r in rows if r not in main})
print("  other kernels in the trace:", "; ".join(others[:8]))
