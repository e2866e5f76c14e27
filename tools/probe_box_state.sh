#!/bin/bash
# Clocks, power and limits of the GPU around an MLA decode run: the same library measures 185 us on
# some boxes and 215-225 us on others (256 rows x 4096 tokens, 16 heads), while the paged decode
# kernel runs the same on both.  Printed side by side so that a slow box can be told from a fast one.
echo "== before"
rocm-smi --showclocks --showpower --showmaxpower --showperflevel --showmemuse 2>&1 | grep -v "^$\|=====" | head -40
echo "== MLA decode, 16 heads x 256 rows x 4096 tokens; paged decode at the metric shape"
MLA_TARGETS=0,0,0 timeout -k 10 120 python3 tools/bench_mla.py 256 16 4096 2>/dev/null | tail -3
(timeout -k 10 100 python3 tools/bench_mla.py 256 16 4096 > /dev/null 2>&1 &) ; sleep 6
echo "== during an MLA run"
rocm-smi --showclocks --showpower 2>&1 | grep -v "^$\|=====" | head -20
wait
timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 100 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*' | head -2
