"""Data-parallel / batch-parallel product path over the REAL backend ("nccl" = RCCL): every rank a child process started with
torch.distributed.run, as the driver starts bench.py.

* world size 1 on one GPU (always runs on the GPU box): `DataParallel.train_step` (building blocks + RCCL collectives between
  them) against the single-process in-library step, `predict` / `evaluate` / `fit` through the collective code;
* world size 2 when two GPUs are visible: the same, each rank on its shard of whole graphs - gradients, loss, k and BatchNorm
  moving statistics of the SINGLE-PROCESS step on the whole batch."""
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
    from gnnkeras_amd.load_MUTAG import load_graphs
    from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
    from gnnkeras_amd.Models.GNN import GNNgraphBased
    from gnnkeras_amd.Models.training import LoopTrainer, Adam
    from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
    from gnnkeras_amd.data_parallel import DataParallel
    from oracle.harness import rel_err
    graphs = load_graphs(limit=96)
    d = 16
    def build():
        inp, lay = get_inout_dims('state', 14, 3, 2, 'g', d); ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0, device=dev)
        ns.set_weights([w * 0.5 if w.ndim == 2 else w for w in ns.get_weights()])
        inp, lay = get_inout_dims('output', 14, 3, 2, 'g', d, hidden_units=[8]); no = MLP(inp[0], lay, ['tanh', 'softmax'], 'glorot_normal', 'glorot_normal', rng=1, device=dev)
        m = GNNgraphBased(ns, no, d, 6, 0.02)
        m.compile(optimizer=Adam(0.01), loss='categorical_crossentropy', metrics=['accuracy'])
        return m
    seq = MultiGraphSequencer([g.copy() for g in graphs], 'g', 'average', 32, shuffle=False, device=dev)
    sizes = [g.nodes.shape[0] for g in graphs[:32]]
    s0 = np.random.default_rng(1).normal(0, 0.1, (sum(sizes), d)).astype(np.float32)
    # single-process step on the whole batch (the in-library gnn_train_step), on every rank
    ref_model = build()
    tr = LoopTrainer(ref_model)
    x, y, sw = seq[0]
    ref = tr.train_step(x, y, sw, state0=torch.from_numpy(s0).to(dev), apply=False)
    ref_grads = [g.cpu().numpy().copy() for g in tr.gs.gradients() + tr.go.gradients()]
    ref_moving = [w.copy() for w in ref_model.net_state.get_weights()[2:4] + ref_model.net_output.get_weights()[2:4]]
    # data-parallel step: this rank's shard
    model = build()
    dpm = DataParallel(model)
    shard = dpm.shard(seq, 0)
    lo, hi = 32 * rank // world, 32 * (rank + 1) // world
    n0 = sum(sizes[:lo]); n1 = n0 + sum(sizes[lo:hi])
    res = dpm.train_step(shard, state0=torch.from_numpy(s0[n0:n1]).to(dev), apply=False)
    got = [g.cpu().numpy() for g in dpm._trainer.gs.gradients() + dpm._trainer.go.gradients()]
    assert res['k'] == ref['k'], (res['k'], ref['k'])
    assert abs(float(res['loss']) - float(ref['loss'])) <= 1e-5 * max(1.0, abs(float(ref['loss'])))
    worst = 0.0
    scale = max(float(np.max(np.abs(r))) for r in ref_grads)
    for g, r in zip(got, ref_grads):
        e = float(np.max(np.abs(g - r)) / max(float(np.max(np.abs(r))), 1e-12))
        assert e <= 2e-4 or float(np.max(np.abs(g - r))) <= 2e-4 * scale, e
        worst = max(worst, min(e, float(np.max(np.abs(g - r))) / scale))
    for a, b in zip(model.net_state.get_weights()[2:4] + model.net_output.get_weights()[2:4], ref_moving):
        assert rel_err(a, b) <= 1e-5
    # inference: the collective predict / evaluate against the single-process calls (state_vect_dim = 0: the forward draws no
    # random state_0, so two calls are comparable)
    inp, lay = get_inout_dims('state', 14, 3, 2, 'g', 0); ns0 = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0, device=dev)
    inp, lay = get_inout_dims('output', 14, 3, 2, 'g', 0); no0 = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1, device=dev)
    m0 = GNNgraphBased(ns0, no0, 0, 5, 0.01)
    m0.compile(optimizer=Adam(0.01), loss='categorical_crossentropy', metrics=['accuracy'])
    dp0 = DataParallel(m0)
    p_dp, e_dp = dp0.predict(seq), dp0.evaluate(seq, return_dict=True)
    p_1, e_1 = m0.predict(seq), m0.evaluate(seq, return_dict=True)
    assert p_dp.shape == (96, 2) and rel_err(p_dp, p_1) <= 1e-5
    assert abs(e_dp['loss'] - e_1['loss']) <= 1e-5 and abs(e_dp['accuracy'] - e_1['accuracy']) <= 1e-6
    assert dpm.predict(seq).shape == (96, 2)
    # fit: two epochs with reshuffling; every rank ends with the same weights
    seq_t = MultiGraphSequencer([g.copy() for g in graphs], 'g', 'average', 32, shuffle=True, device=dev)
    hist = dpm.fit(seq_t, epochs=2, verbose=0, validation_data=seq)
    w = torch.cat([torch.from_numpy(a.reshape(-1)) for a in model.net_state.get_weights() + model.net_output.get_weights()]).to(dev)
    parts = [torch.empty_like(w) for _ in range(world)]
    dist.all_gather(parts, w)
    assert all(torch.equal(parts[0], p_) for p_ in parts), 'ranks diverged'
    assert len(hist['loss']) == 2 and np.isfinite(hist['loss']).all() and 'val_loss' in hist
    # "replicas" mode: the in-library step (gnn_train_step, persistent kernels) on the own shard + one weighted all-reduce over RCCL;
    # at world size 1 it is the single-process step; at any size every rank ends an epoch with the same weights
    model_r = build()
    dpr = DataParallel(model_r, exact=False)
    res_r = dpr.train_step(dpr.shard(seq, 0), state0=torch.from_numpy(s0[n0:n1]).to(dev), apply=False)
    assert dpr._trainer.dp is None
    if world == 1:
        got_r = [g.cpu().numpy() for g in dpr._trainer.gs.gradients() + dpr._trainer.go.gradients()]
        assert res_r['k'] == ref['k'] and abs(float(res_r['loss']) - float(ref['loss'])) <= 1e-6
        for g, r in zip(got_r, ref_grads): assert np.max(np.abs(g - r)) <= 1e-6 * max(1.0, float(np.max(np.abs(r))))
    hist_r = dpr.fit(MultiGraphSequencer([g.copy() for g in graphs], 'g', 'average', 32, shuffle=True, device=dev), epochs=2, verbose=0)
    w = torch.cat([torch.from_numpy(a.reshape(-1)) for a in model_r.net_state.get_weights() + model_r.net_output.get_weights()]).to(dev)
    parts = [torch.empty_like(w) for _ in range(world)]
    dist.all_gather(parts, w)
    assert all(torch.equal(parts[0], p_) for p_ in parts), 'replicas diverged'
    assert np.isfinite(hist_r['loss']).all()
    # heterogeneous models (round 4): the replica mode runs the in-library composite step (csrc/train_composite.hpp) on every shard
    from gnnkeras_amd.synth import er_composite_graph
    from gnnkeras_amd.Models.CompositeGNN import CompositeGNNnodeBased
    from gnnkeras_amd.Sequencers.GraphSequencers import CompositeMultiGraphSequencer
    dims = (5, 3, 2)
    cgraphs = [er_composite_graph(30 + i, 90 + 2 * i, dim_node_label=dims, seed=50 + i) for i in range(16)]
    def cbuild():
        inp, lay = get_inout_dims('state', dims, 3, 2, 'n', 8)
        nsc = [MLP(i, lay, 'tanh', 'lecun_normal', 'lecun_normal', rng=t, device=dev) for t, i in enumerate(inp)]
        inp, lay = get_inout_dims('output', dims, 3, 2, 'n', 8); noc = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=7, device=dev)
        mc = CompositeGNNnodeBased(nsc, noc, 8, 4, 0.0)
        mc.compile(optimizer=Adam(0.01), loss='categorical_crossentropy', metrics=['accuracy'])
        return mc
    cseq = CompositeMultiGraphSequencer(cgraphs, 'n', 'average', 8, shuffle=False, device=dev)
    # ... and since round 5 the EXACT mode takes them too (per-type statistics combined over the ranks with the shard's row count of the
    # type as weight - zero included; tests/test_data_parallel.py runs it at world 2 / 3 over gloo): here the product path over RCCL
    mx = cbuild()
    dpx = DataParallel(mx, exact=True)
    csz_x = [g.nodes.shape[0] for g in cgraphs[:8]]
    xlo, xhi = 8 * rank // world, 8 * (rank + 1) // world
    x0 = sum(csz_x[:xlo]); x1_ = x0 + sum(csz_x[xlo:xhi])
    xs0 = np.random.default_rng(3).normal(0, 0.1, (sum(csz_x), 8)).astype(np.float32)
    res_x = dpx.train_step(dpx.shard(cseq, 0), state0=torch.from_numpy(xs0[x0:x1_]).to(dev), apply=False)
    assert not dpx._trainer._native_step_applies(cseq[0][1]) and res_x['k'] == 4
    mref = cbuild(); tref = LoopTrainer(mref)
    xr, yr, swr = cseq[0]
    rref = tref.train_step(xr, yr, swr, state0=torch.from_numpy(xs0).to(dev), apply=False)
    gref = [g for t_ in tref.gs for g in t_.gradients()] + tref.go.gradients()
    gx = [g for t_ in dpx._trainer.gs for g in t_.gradients()] + dpx._trainer.go.gradients()
    assert abs(float(res_x['loss']) - float(rref['loss'])) <= 1e-5
    sc = max(float(b_.abs().max()) for b_ in gref)
    for a_, b_ in zip(gx, gref): assert float((a_ - b_).abs().max()) <= 2e-5 * max(float(b_.abs().max()), sc), 'exact composite step'
    mc = cbuild()
    dpc = DataParallel(mc, exact=False)
    csz = [g.nodes.shape[0] for g in cgraphs[:8]]
    clo, chi = 8 * rank // world, 8 * (rank + 1) // world
    c0 = sum(csz[:clo]); c1 = c0 + sum(csz[clo:chi])
    cs0 = np.random.default_rng(3).normal(0, 0.1, (sum(csz), 8)).astype(np.float32)
    res_c = dpc.train_step(dpc.shard(cseq, 0), state0=torch.from_numpy(cs0[c0:c1]).to(dev), apply=False)
    assert dpc._trainer._native_step_applies(cseq[0][1]) and res_c['k'] == 4
    if world == 1:          # the single-process composite step on the whole batch
        m1 = cbuild(); t1 = LoopTrainer(m1)
        x1, y1, sw1 = cseq[0]
        r1 = t1.train_step(x1, y1, sw1, state0=torch.from_numpy(cs0).to(dev), apply=False)
        g1 = [g for t_ in t1.gs for g in t_.gradients()] + t1.go.gradients()
        gc = [g for t_ in dpc._trainer.gs for g in t_.gradients()] + dpc._trainer.go.gradients()
        assert abs(float(res_c['loss']) - float(r1['loss'])) <= 1e-6
        for a_, b_ in zip(gc, g1): assert float((a_ - b_).abs().max()) <= 1e-6 * max(1.0, float(b_.abs().max()))
    hist_c = dpc.fit(CompositeMultiGraphSequencer(cgraphs, 'n', 'average', 8, shuffle=True, device=dev), epochs=2, verbose=0)
    wts = [a_ for n_ in mc.net_state for a_ in n_.get_weights()] + mc.net_output.get_weights()
    w = torch.cat([torch.from_numpy(a_.reshape(-1)) for a_ in wts]).to(dev)
    parts = [torch.empty_like(w) for _ in range(world)]
    dist.all_gather(parts, w)
    assert all(torch.equal(parts[0], p_) for p_ in parts), 'composite replicas diverged'
    assert np.isfinite(hist_c['loss']).all()
    dist.barrier(); torch.cuda.synchronize()
    dist.destroy_process_group()
    if rank == 0: print('DP_OK ' + json.dumps({'k': res['k'], 'worst_grad_err': worst, 'loss': [float(v) for v in hist['loss']]}))
''')


def _run_ranks(tmp_path, world):
    script = tmp_path / 'dp_worker.py'
    script.write_text(WORKER)
    env = dict(os.environ, GNN_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}', '--master-addr', '127.0.0.1',
           '--master-port', str(29640 + world), str(script)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert res.returncode == 0 and 'DP_OK' in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]


def test_data_parallel_over_rccl_world_1(tmp_path):
    _run_ranks(tmp_path, 1)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs at least 2 GPUs on the box')
def test_data_parallel_over_rccl_world_2(tmp_path):
    _run_ranks(tmp_path, 2)
