#!/bin/bash
# Kernel-trace split of an MLA decode call into its kernels (attention kernel, merge kernel):
# tools/trace_mla_parts.sh "rows heads tokens" ...   (run on the GPU box from the repository root)
export TMPDIR=/tmp
root=$PWD
for shape in "$@"; do
  rm -rf /tmp/mla_parts
  (cd /tmp && MLA_TARGETS=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mla_parts -o t -- python3 $root/tools/bench_mla.py $shape > /tmp/mla_parts.out 2>&1) || { tail -3 /tmp/mla_parts.out; exit 1; }
  grep mla_decode /tmp/mla_parts.out | sed 's/R=512 P=64//; s/GB\/s.*TB\/s)//'
  python3 - <<PY
import csv, glob
f = glob.glob('/tmp/mla_parts/**/t_kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'mla' in r['Name']:
        name = r['Name'].split('<')[0].split('::')[-1][:40]
        print(f"    {name:42s} {float(r['AverageNs']) / 1000:7.1f} us avg over {r['Calls']} launches")
PY
done
