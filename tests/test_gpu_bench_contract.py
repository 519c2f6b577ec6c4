"""bench.py keeps its contract: ONE JSON line on stdout with the agreed keys, a roofline object for the dominant kernel
(frac = achieved / peak, achieved from live HIP events) and, when asked, the CPU baseline."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra):
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1', *extra],
                         capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_small_workload():
    d = run_bench('--workload', 'c3', '--no-mutag')
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in d, key
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1 and d['higher_is_better'] is True
    assert d['dtype'] == 'f32' and d['data'] == 'synthetic' and d['vs_baseline'] is None and 'workload' in d['config']
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and 0.3 < r['frac'] < 1.0
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] >= 1 and c['value'] > 0 and 'sample' in c
    assert d['value'] > 100 * c['value']                       # the device loop is orders of magnitude ahead of the host
    assert abs(d['value'] - 1e6 * 50 * 2 / (d['ms_per_step'] * 2e-3)) / d['value'] < 1e-6      # arcs x k x steps / time
