"""Copy the summaries of one tools/collect_headline.sh run from gpurun_out/<tag>/ into profiles/
(tracked):   python tools/publish_headline.py r2"""
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, 'gpurun_out', tag)
dst = os.path.join(ROOT, 'profiles')


def put(rel, name):
    found = glob.glob(os.path.join(src, rel))
    if not found:
        print('missing', rel)
        return
    shutil.copyfile(found[0], os.path.join(dst, f'{tag}_{name}'))


put('bench_line.json', 'bench_line.json')
put('bench_line_steps20.json', 'bench_line_steps20.json')
put('bench_line_c2.json', 'bench_line_c2.json')
put('library_sha256.txt', 'library_sha256.txt')
put('pmc_traffic_metric.json', 'pmc_traffic.json')
put('pmc_traffic_c2.json', 'pmc_traffic_c2.json')
for cfg, short in (('metric', 'bench'), ('c2', 'c2')):
    put(f'trace_{cfg}/**/{cfg}_kernel_stats.csv', f'{short}_kernel_stats.csv')
    put(f'trace_{cfg}/{cfg}_kernel_stats.csv', f'{short}_kernel_stats.csv')
    put(f'bench_line_{cfg}_under_rocprof.json', f'bench_line_{cfg}_under_rocprof.json')
    put(f'pmc_fetch_{cfg}/**/*counter_collection.csv', f'pmc_fetch_size_{cfg}.csv')
    put(f'pmc_fetch_{cfg}/*counter_collection.csv', f'pmc_fetch_size_{cfg}.csv')
    put(f'pmc_write_{cfg}/**/*counter_collection.csv', f'pmc_write_size_{cfg}.csv')
    put(f'pmc_write_{cfg}/*counter_collection.csv', f'pmc_write_size_{cfg}.csv')
print(sorted(f for f in os.listdir(dst) if f.startswith(tag + '_')))
