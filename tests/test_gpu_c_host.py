"""The C ABI used from plain C (examples/c_host.c): gcc builds a host program that links libgnnloop.so and the HIP
runtime, runs the node-focused loop on a small graph through every iteration path and checks (k, state, out) against
its own scalar double-precision restatement of the reference recurrence."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'gnnkeras_amd', 'csrc')


def build(tmp_path):
    exe = str(tmp_path / 'c_host')
    cmd = ['gcc', '-std=c99', '-O1', '-D__HIP_PLATFORM_AMD__', '-I/opt/rocm/include', '-I' + os.path.join(ROOT, 'include'),
           os.path.join(ROOT, 'examples', 'c_host.c'), '-L' + CSRC, '-lgnnloop', '-L/opt/rocm/lib', '-lamdhip64',
           '-Wl,-rpath,' + CSRC, '-Wl,-rpath,/opt/rocm/lib', '-lm', '-o', exe]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return exe


def test_c_host_builds_as_c99(tmp_path):
    """No GPU needed: the header is valid C and the library resolves every symbol the C host uses."""
    if not os.path.exists(os.path.join(CSRC, 'libgnnloop.so')):
        pytest.skip('libgnnloop.so not built')
    build(tmp_path)


@pytest.mark.gpu
def test_c_host_runs_and_matches_its_own_restatement(tmp_path):
    exe = build(tmp_path)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    sys.stdout.write(res.stdout)
    assert res.returncode == 0, res.stdout + res.stderr
    assert 'c_host: OK' in res.stdout
    assert res.stdout.count(' ok') == 4          # default, un-fused, one launch per iteration, whole loop in one launch
