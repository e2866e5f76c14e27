"""Copy the summaries of one tools/collect_profiles.sh run from gpurun_out/<tag>/ into profiles/
(tracked), deriving the small JSON digests:   python tools/publish_profiles.py r1e"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, 'gpurun_out', tag)
dst = os.path.join(ROOT, 'profiles')


def put(rel, name):
    shutil.copyfile(os.path.join(src, rel), os.path.join(dst, f'{tag}_{name}'))


subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'publish_headline.py'), tag])
with open(os.path.join(dst, f'{tag}_secondary_benchmarks.txt'), 'w') as f:      # (collected in one or two GPU calls)
    for part in ('secondary.txt', 'secondary_a.txt', 'secondary_b.txt'):
        if os.path.exists(os.path.join(src, part)):
            f.write(open(os.path.join(src, part)).read())
put('pmc_mla_fetch/fetch_counter_collection.csv', 'pmc_mla_fetch_size.csv')
for name in ('gqa', 'mla', 'mla_wide', 'prefill', 'engine', 'prefill_f32', 'c2tune'):
    put(f'trace_{name}/{name}_kernel_stats.csv', f'{name}_kernel_stats.csv')
for rel, name in [('engine_decode_step_kernels.txt', 'engine_decode_step_kernels.txt'),
                  ('sampler_phases.txt', 'sampler_phases.txt'),
                  ('mla_wide_phases.txt', 'mla_wide_phases.txt'),
                  ('trace_exchange/exchange_kernel_stats.csv', 'exchange_kernel_stats.csv'),
                  ('bench_line_exchange_under_rocprof.json', 'bench_line_exchange_under_rocprof.json')]:
    if os.path.exists(os.path.join(src, rel)):
        put(rel, name)
put('trace_paged_prefill/pp_kernel_stats.csv', 'paged_prefill_kernel_stats.csv')


def counters(path, kernel):
    acc = collections.defaultdict(list)
    for f in glob.glob(path):
        for r in csv.DictReader(open(f)):
            if kernel in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in acc.items()}, max((len(v) for v in acc.values()), default=0)


# MLA read traffic against the algorithmic bytes (256 rows x 4096 tokens x 576 elements x 2 B + queries/outputs)
c, n = counters(os.path.join(src, 'pmc_mla_fetch/fetch_counter_collection.csv'), 'mla_decode_kernel')
rows, ctx, heads, R, P = 256, 4096, 16, 512, 64
alg = rows * ctx * (R + P) * 2 + rows * heads * (R + P) * 2 + rows * heads * R * 2 + rows * (ctx // 16) * 4
json.dump({"kernel": "mla_decode_kernel_w4<BF16,512,64>",
           "workload": "256 rows x 4096 tokens, 16 heads, kv_lora_rank 512, rope 64 (tools/bench_mla.py 256 16 4096)",
           "launches": n, "FETCH_SIZE_KiB_avg": c['FETCH_SIZE'],
           "hbm_read_bytes_per_launch (2 x FETCH_SIZE x 1024, gfx950 correction)": int(2 * c['FETCH_SIZE'] * 1024),
           "algorithmic_bytes_per_launch": alg, "ratio": round(2 * c['FETCH_SIZE'] * 1024 / alg, 4)},
          open(os.path.join(dst, f'{tag}_pmc_mla_traffic.json'), 'w'), indent=1)

# prompt-attention kernel: issue, LDS and matrix-pipe activity
for shape, label in (('d128', '4 sequences x 8192 tokens, 32/8 heads, D=128, bf16 (tools/bench_prefill.py 4 8192 32 8 128)'),
                     ('d64', '2 sequences x 16384 tokens, 8 heads, D=64, bf16 (tools/bench_prefill.py 2 16384 8 8 64)')):
    c, n = counters(os.path.join(src, f'pmc_prefill_{shape}/*/*counter_collection.csv'), 'prefill_mfma')
    cycles = c['GRBM_GUI_ACTIVE'] / 8
    simd_cycles = cycles * 1024
    json.dump({"kernel": "prefill_mfma_kernel", "workload": label, "launches": n,
               "counters (average per launch)": {k: round(v) for k, v in sorted(c.items())},
               "kernel_cycles (GRBM_GUI_ACTIVE summed over 8 XCDs / 8)": round(cycles),
               "mfma_busy_fraction (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles))":
                   round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles, 4),
               "valu_active_fraction (4 x SQ_ACTIVE_INST_VALU / SIMD cycles; counts quad-cycles, MFMA issue included)":
                   round(4 * c['SQ_ACTIVE_INST_VALU'] / simd_cycles, 4),
               "lds_busy_fraction (SQ_LDS_IDX_ACTIVE / (256 CUs x kernel cycles))":
                   round(c['SQ_LDS_IDX_ACTIVE'] / (256 * cycles), 4),
               "lds_conflict_fraction (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)":
                   round(c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE'], 4),
               "waves_per_simd (4 x SQ_WAVE_CYCLES / SIMD cycles)": round(4 * c['SQ_WAVE_CYCLES'] / simd_cycles, 2)},
              open(os.path.join(dst, f'{tag}_pmc_prefill_{shape}.json'), 'w'), indent=1)
c, n = counters(os.path.join(src, 'pmc_prefill_f32/*/*counter_collection.csv'), 'prefill_mfma_f32')
if c:
    cycles = c['GRBM_GUI_ACTIVE'] / 8
    json.dump({"kernel": "prefill_mfma_f32_kernel<64, false>",
               "workload": "16 sequences x 2048 tokens, 8 heads, D=64, fp32 (tools/bench_prefill_fp32.py 16 2048 8 8 64)",
               "launches": n, "counters (average per launch)": {k: round(v) for k, v in sorted(c.items())},
               "kernel_cycles (GRBM_GUI_ACTIVE summed over 8 XCDs / 8)": round(cycles),
               "mfma_busy_fraction (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles))":
                   round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (cycles * 1024), 4),
               "lds_busy_fraction": round(c['SQ_LDS_IDX_ACTIVE'] / (256 * cycles), 4),
               "lds_conflict_fraction": round(c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE'], 4),
               "waves_per_simd (4 x SQ_WAVE_CYCLES / SIMD cycles)": round(4 * c['SQ_WAVE_CYCLES'] / (cycles * 1024), 2)},
              open(os.path.join(dst, f'{tag}_pmc_prefill_f32.json'), 'w'), indent=1)
c, n = counters(os.path.join(src, 'pmc_mla_mh/*/*counter_collection.csv'), 'mla_decode_kernel_mh')
if c:
    cycles = c['GRBM_GUI_ACTIVE'] / 8
    json.dump({"kernel": "mla_decode_kernel_mh<BF16,512,64>",
               "workload": "128 rows x 4096 tokens, 128 heads, kv_lora_rank 512, rope 64 (tools/bench_mla.py 128 128 4096)",
               "launches": n, "counters (average per launch)": {k: round(v) for k, v in sorted(c.items())},
               "kernel_cycles (GRBM_GUI_ACTIVE summed over 8 XCDs / 8)": round(cycles),
               "mfma_busy_fraction": round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (cycles * 1024), 4),
               "lds_busy_fraction": round(c['SQ_LDS_IDX_ACTIVE'] / (256 * cycles), 4),
               "lds_conflict_fraction": round(c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE'], 4),
               "wave_wait_fraction (SQ_WAIT_ANY / SQ_WAVE_CYCLES)": round(c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'], 4)},
              open(os.path.join(dst, f'{tag}_pmc_mla_many_heads.json'), 'w'), indent=1)
c, n = counters(os.path.join(src, 'pmc_mla_wide/*/*counter_collection.csv'), 'mla_decode_kernel_wide')
if c:
    cycles = c['GRBM_GUI_ACTIVE'] / 8
    json.dump({"kernel": "mla_decode_kernel_wide<BF16, dense>",
               "workload": "256 rows x 4096 tokens, 128 heads, kv_lora_rank 512, rope 64 (tools/bench_mla.py 256 128 4096)",
               "launches": n, "counters (average per launch)": {k: round(v) for k, v in sorted(c.items())},
               "kernel_cycles (GRBM_GUI_ACTIVE summed over 8 XCDs / 8)": round(cycles),
               "mfma_busy_fraction": round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (cycles * 1024), 4),
               "lds_busy_fraction": round(c['SQ_LDS_IDX_ACTIVE'] / (256 * cycles), 4),
               "lds_conflict_fraction": round(c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE'], 4),
               "wave_wait_fraction (SQ_WAIT_ANY / SQ_WAVE_CYCLES)": round(c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'], 4)},
              open(os.path.join(dst, f'{tag}_pmc_mla_wide.json'), 'w'), indent=1)
print(sorted(f for f in os.listdir(dst) if f.startswith(tag)))
