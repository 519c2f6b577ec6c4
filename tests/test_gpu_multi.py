"""The sharded loop's product path over the REAL backend ("nccl" = RCCL), every rank a child process started with
torch.distributed.run exactly as the driver starts bench.py:

* world size 1 on one GPU (always runs on the GPU box): the overlapped forward — async collective, own-range partial,
  stream-ordered wait — through RCCL's API for both exchange layouts, against the oracle;
* world size 2 when at least two GPUs are visible (skipped otherwise): ShardedLoop and HaloShardedLoop, overlap on and off,
  against the single-process oracle, plus `bench.py --gpus 2` in the driver's own command form (self-launch).
"""
import json
import os
import subprocess
import sys
import textwrap

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

WORKER = textwrap.dedent('''
    import os, sys, json
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, os.environ['GNN_ROOT'])
    rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ['LOCAL_RANK'])
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    from gnnkeras_amd import GraphObject
    from gnnkeras_amd.synth import er_graph
    from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
    from gnnkeras_amd.Models.GNN import GNNnodeBased
    from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
    from gnnkeras_amd.distributed import make_sharded_loop
    from oracle.harness import oracle_loop, rel_err
    N, E, d = 40_003, 400_000, 64
    g = er_graph(N, E, seed=7)
    inp, lay = get_inout_dims('state', 14, 3, 2, 'n', d); ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0, device=dev)
    ns.set_weights([w * 0.3 if w.ndim == 2 else w for w in ns.get_weights()])
    inp, lay = get_inout_dims('output', 14, 3, 2, 'n', d); no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1, device=dev)
    s0 = np.random.default_rng(1).normal(0, 0.1, (N, d)).astype(np.float32)
    report = {}
    for threshold in (0.0, 0.02):
        model = GNNnodeBased(ns, no, d, 6, threshold)
        x = MultiGraphSequencer([g], 'n', 'average', 1, shuffle=False, device=dev)[0][0]
        k64, st64, o64 = oracle_loop(model, x, s0, np.float64, exact_order=False)
        for exchange in ('allgather', 'direct', 'halo', 'auto'):
            for overlap in (False, True):
                sl = make_sharded_loop(model, g, rank, world, dev, exchange=exchange, overlap=overlap)
                assert sl.overlap == overlap, 'split refused'
                if exchange == 'auto' and world > 1: assert sl.transport in ('ring', 'direct') and len(sl.transport_times) == 2
                k, st, o = sl.forward(torch.from_numpy(s0).to(dev))
                torch.cuda.synchronize()
                lo, hi = sl.plan.lo, sl.plan.hi
                idx = sl.plan.out_index + lo
                es, eo = rel_err(st.cpu().numpy(), st64[lo:hi]), rel_err(o.cpu().numpy(), o64[np.isin(np.flatnonzero(np.ones(N, bool)), idx)])
                assert float(k) == float(k64), (float(k), k64)
                assert es <= 1e-5 and eo <= 1e-5, (exchange, overlap, es, eo)
                report[f'{threshold}/{exchange}/{overlap}'] = [float(k), es, eo]
                if threshold == 0.0:
                    prof = sl.profile_iteration(torch.from_numpy(s0).to(dev), reps=3)
                    assert prof['kernel_s'] > 0 and prof['iteration_s'] > 0
                # the loop driven from native code (csrc/shard_loop.hpp: one C call for all iterations, the exchange over the RCCL C API on
                # the library's own communicator - also at world size 1, where it is an all-gather of the one slice): the SAME BITS
                if exchange in ('allgather', 'direct'):
                    for chunks in ((1, 2) if overlap else (1,)):
                        if chunks > 1 and sl.set_pipeline(chunks) != chunks: continue
                        kp, stp, op = [t.clone() for t in sl.forward(torch.from_numpy(s0).to(dev))]
                        sl.enable_native_loop(with_comm=True)
                        kn, stn, on = sl.forward(torch.from_numpy(s0).to(dev))
                        torch.cuda.synchronize()
                        assert float(kn) == float(kp) == float(k64) and torch.equal(stn, stp) and torch.equal(on, op), (exchange, overlap, chunks)
                        sl.native_loop = False
                        report[f'{threshold}/{exchange}/{overlap}/native/{chunks}'] = 'bit-identical'
                    sl.close()
    dist.barrier(); torch.cuda.synchronize()
    dist.destroy_process_group()
    if rank == 0: print('MULTI_OK ' + json.dumps(report))
''')


def _run_ranks(tmp_path, world):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    env = dict(os.environ, GNN_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}', '--master-addr', '127.0.0.1',
           '--master-port', str(29600 + world), str(script)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert res.returncode == 0 and 'MULTI_OK' in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]


def test_overlapped_forward_over_rccl_world_1(tmp_path):
    _run_ranks(tmp_path, 1)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs at least 2 GPUs on the box')
def test_sharded_loops_over_rccl_world_2(tmp_path):
    _run_ranks(tmp_path, 2)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs at least 2 GPUs on the box')
def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset (the driver's command form) starts its own ranks and prints ONE
    JSON line with n_gpus = 2, a roofline record and the exchange / kernel split."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--workload', 'c3'],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r['n_gpus'] == 2 and r['value'] > 0 and r['roofline']['frac'] > 0
    assert set(r['per_iteration_ms']) >= {'kernel', 'exchange', 'iteration_overlapped'}


def test_bench_forced_sharded_path_on_one_gpu():
    """The N > 1 code path of bench.py (ShardedLoop + RCCL collectives + overlap) with the one rank a 1-GPU box allows."""
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29617')
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1', '--workload', 'c3',
                          '--force-sharded'], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    r = json.loads([l for l in res.stdout.splitlines() if l.startswith('{')][0])
    assert r['n_gpus'] == 1 and 'per_iteration_ms' in r and r['roofline']['kernel'].startswith('k_state_fused4')
