"""Short soak of the in-launch hand-offs (scripts/soak.py does the long run): the persistent whole-loop kernel must match
the one-launch-per-iteration kernel bit for bit on every MUTAG batch, and repeated runs of the wave-specialised kernel
must be bitwise reproducible."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_soak_short():
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'soak.py'), '2'], capture_output=True, text=True, timeout=900)
    sys.stdout.write(res.stdout[-2000:])
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert 'soak: OK' in res.stdout
