#!/bin/bash
# A/B of heads-per-workgroup plans for the small decode config and GQA shapes.
set -e
for hpw in 8 4 2 1 0; do
  export MI355_PAGED_HPW=$hpw
  echo "== MI355_PAGED_HPW=$hpw"
  timeout -k 10 120 python tools/tune_decode.py c2 256 512
done
unset MI355_PAGED_HPW
timeout -k 10 120 python tools/tune_decode.py metric 256
timeout -k 10 120 python tools/bench_gqa.py
