"""bench.py prints ONE JSON line with the fields the benchmark contract names (needs an MI355X).

Each case runs bench.py as a child process, as the driver does, with a handful of steps."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '6', '--warmup', '2', *flags],
                          capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [l for l in proc.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f"stdout must hold the one JSON line, got {len(lines)} lines"
    return json.loads(lines[0])


def _check_common(line, steps=6, warmup=2):
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
        assert key in line, key
    assert line['n_gpus'] == 1 and line['steps'] == steps and line['warmup'] == warmup
    assert line['unit'] == 'tokens/s' and line['higher_is_better'] is True and line['vs_baseline'] is None
    assert line['dtype'] == 'bf16' and line['data'] == 'synthetic' and 'workload' in line['config']
    assert 'model' not in line['config']
    r = line['roofline']
    for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert key in r, key
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3 and 0.3 < r['frac'] < 1.0
    # tokens/s and ms per step describe the same run
    batch = line['config']['batch_per_gpu']
    assert abs(line['value'] - batch / (line['ms_per_step'] * 1e-3)) / line['value'] < 0.01
    # the launch the roofline prices fits inside a step
    assert r['avg_launch_ms'] < line['ms_per_step']


def test_default_line_is_the_metric_config_with_a_cpu_baseline():
    line = _run()
    _check_common(line)
    assert line['metric'] == 'decode tokens/sec + achieved HBM GB/s, block_size=16 batch=256 seq=2048'
    assert line['scaling'] == 'weak' and line['config']['batch_per_gpu'] == 256 and line['config']['context'] == 2048
    cpu = line['cpu_baseline']
    assert cpu['kind'] == 'port' and cpu['unit'] == 'tokens/s' and cpu['value'] > 0 and cpu['cores'] >= 1 and cpu['sample']
    assert line['roofline']['frac'] > 0.5      # (a gross-regression guard, not the measurement: that is bench.py's own)


def test_config_2_and_the_fused_step():
    two = _run('--config', 'c2', '--no-cpu-baseline')
    _check_common(two)
    assert two['config']['batch_per_gpu'] == 64 and two['config']['context'] == 1024 and 'cpu_baseline' not in two
    fused = _run('--config', 'c2', '--no-cpu-baseline', '--fused-step')
    _check_common(fused)
    assert 'one launch' in fused['config']['workload'] and 'append' in fused['roofline']['kernel']
    assert fused['roofline']['traffic'] is None          # the counters on file belong to the two-launch step's kernel


def test_fixed_total_batch():
    line = _run('--strong', '512', '--no-cpu-baseline')
    _check_common(line)
    assert line['scaling'] == 'strong' and line['config']['batch_per_gpu'] == 512
