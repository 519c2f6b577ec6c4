"""The exchange inside the iteration kernel (SURVEY 8e "each rank writes its slice to all peers, 1 hop"): two PROCESSES on
the one GPU of the box, each owning half of the node range, their full state buffers mapped into each other with hipIpcOpenMemHandle; the
wave-specialised kernel's epilogue stores every new row to both.  Functional validation only (bit-identity with the slice-copy exchange): IPC
between two processes on one device exercises the mappings, the arrival words and the ordering - not the links.  tests/peer_worker.py is a rank."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('d,thr', [(64, 0.0), (32, 0.05)])
def test_two_processes_exchange_state_rows_through_peer_stores(d, thr):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), PEER_D=str(d), PEER_THR=str(thr),
                   HSA_ENABLE_IPC_MODE_LEGACY='0', PYTHONPATH=root + os.pathsep + os.environ.get('PYTHONPATH', ''))
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, 'tests', 'peer_worker.py')], env=env, cwd=root, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try: o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs: q.kill()
            raise
        outs.append(o)
    assert all(p.returncode == 0 for p in procs), '\n----\n'.join(o[-3000:] for o in outs)
    assert 'PEER_OK' in outs[0], outs[0][-3000:]
