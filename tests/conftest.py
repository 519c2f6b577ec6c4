import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'graph_fixtures.npz'))


@pytest.fixture(scope='session')
def mutag_graphs():
    from gnnkeras_amd.load_MUTAG import load_graphs
    return load_graphs()


def pytest_collection_finish(session):
    """GPU sessions that contain the deep parity tests (tests/test_gpu_round4.py: BASELINE C3 / C4 / C5 for the 50 iterations the bench
    times) start those tests' float64 oracle runs NOW, on worker threads: 25 s .. 3 min of mostly single-threaded NumPy each, which then
    overlap the minutes of other tests in front of them instead of being waited for."""
    names = [it.name for it in session.items if 'test_gpu_round4' in str(getattr(it, 'fspath', ''))]
    if not names or len(session.items) < 40: return          # (a hand-picked run: the deep test starts its own job)
    try:
        import torch
        if not torch.cuda.is_available(): return
        mod = sys.modules.get('test_gpu_round4')
        if mod is not None: mod.start_deep_oracles(names)
        mod5 = sys.modules.get('test_gpu_round5')          # (the million-node train-step oracle: tests/test_gpu_round5.py)
        if mod5 is not None: mod5.start_train_oracle([it.name for it in session.items])
        mod6 = sys.modules.get('test_gpu_round6')          # (the C5-size heterogeneous train-step oracle: tests/test_gpu_round6.py)
        if mod6 is not None: mod6.start_c5_train_oracle([it.name for it in session.items])
        # the float64 autograd oracles of the large-graph training tests (tests/test_gpu_training.py: prefetch_oracle), a few at a time
        modt = sys.modules.get('test_gpu_training')
        jobs = [(it.function, dict(it.callspec.params)) for it in session.items
                if getattr(getattr(it, 'function', None), '_prefetch_oracle', False) and hasattr(it, 'callspec')]
        if modt is not None and jobs:
            import threading, queue
            q = queue.Queue()
            for j in jobs: q.put(j)

            def worker():
                while True:
                    try: fn, kw = q.get_nowait()
                    except queue.Empty: return
                    modt.run_oracle_only(fn, kw)
            for _ in range(4): threading.Thread(target=worker, daemon=True).start()
    except Exception:
        pass                                                   # (never fail a collection over a head start)
