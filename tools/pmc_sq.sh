#!/bin/bash
# SQ counters (issue, waits, LDS conflicts, MFMA busy) of one kernel, averaged over its launches;
# one rocprofv3 --pmc pass per counter group.
#   tools/pmc_sq.sh OUTDIR KERNEL_NAME_SUBSTRING tools/bench_prefill.py 4 8192 32 8 128
OUT=$1; KERNEL=$2; shift 2
mkdir -p "$OUT"
export BENCH_MFMA_ONLY=1
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - > /dev/null
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY \
  --output-format csv -d "$OUT/a" -o a -- python3 "$@" > /dev/null 2> "$OUT/a.log" || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE \
  --output-format csv -d "$OUT/b" -o b -- python3 "$@" > /dev/null 2> "$OUT/b.log" || exit 1
python3 - "$OUT" "$KERNEL" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for grp in 'ab':
    for f in glob.glob(f'{out}/{grp}/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if sys.argv[2] in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in sorted(acc.items()):
            print(f'{k:32s} {sum(v)/len(v):16.0f}  (n={len(v)})')
PY
