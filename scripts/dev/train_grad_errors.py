"""Per-tensor errors of one large-graph training step (40 000 nodes, d = 64, BatchNorm, 4 iterations) against torch autograd in float64,
in-library step and Python building blocks.  DBG_ACT=relu|selu|.., DBG_BN=0, DBG_DETAIL=1; GNN_TRAIN_BF16X6=0 / GNN_TRAIN_WGRAD32=0 for the f32 kernels.
(profiles/r04_notes.txt sections 6-7 quote it.)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from gnnkeras_amd import GraphObject
from gnnkeras_amd.synth import er_graph
from gnnkeras_amd.Models.GNN import GNNnodeBased
from gnnkeras_amd.Models.training import LoopTrainer, SGD
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
from test_gpu_training import nets, oracle_step
d, bn, mode = 64, os.environ.get('DBG_BN', '1') == '1', 'average'
ACT = os.environ.get('DBG_ACT', 'selu')
rng = np.random.default_rng(d)
N = 40_000
g = er_graph(N, 6 * N, seed=5, aggregation_mode=mode)
om = rng.random(N) < 0.6
t = np.zeros((int(om.sum()), 2)); t[np.arange(len(t)), rng.integers(0, 2, len(t))] = 1
g = GraphObject(g.nodes, g.arcs, t, focus='n', set_mask=rng.random(N) < 0.9, output_mask=om, aggregation_mode=mode, sample_weight=rng.uniform(0.5, 1.5, len(t)))
x, y, sw = MultiGraphSequencer([g], 'n', mode, 1, shuffle=False)[0]
ns, no = nets('n', d, bn, act=ACT, scale=0.5)
s0 = rng.normal(0, 0.1, (N, d)).astype(np.float32)
model = GNNnodeBased(ns, no, d, 4, 0.0)
model.compile(optimizer=SGD(0.0), loss='categorical_crossentropy')
want = oracle_step(model, x, y, sw, s0, 'categorical_crossentropy', False)
for native in (True, False):
    tr = LoopTrainer(model); tr.use_native_step = native
    res = tr.train_step(x, y, sw, state0=torch.from_numpy(s0).cuda(), apply=False)
    errs = []
    for got, ref in [(tr.gs.gradients(), want['grads_state']), (tr.go.gradients(), want['grads_output'])]:
        for g_, r in zip(got, ref):
            errs.append(float(np.max(np.abs(g_.cpu().numpy() - r)) / max(float(np.max(np.abs(r))), 1e-12)))
    yp = float(np.max(np.abs(res['y_pred'].cpu().numpy() - want['y_pred'])))
    print('native' if native else 'blocks', os.environ.get('GNN_TRAIN_BF16X6', '1'), 'k', res['k'], 'loss err', abs(float(res['loss']) - want['loss']), 'y_pred abs err', yp, 'grad rel errs', ['%.1e' % e for e in errs])
    if native and os.environ.get('DBG_DETAIL'):
        gk = tr.gs.gradients()[2].cpu().numpy(); rk = want['grads_state'][2]
        dk = np.abs(gk - rk)
        print('kernel grad |ref| max', np.abs(rk).max(), 'diff max', dk.max())
        print('per input-row block max diff: state', dk[:64].max(), 'labels', dk[64:78].max(), 'agg', dk[78:142].max(), 'rest', dk[142:].max())
        cols = dk.max(axis=0); print('worst columns', np.argsort(-cols)[:8], cols[np.argsort(-cols)[:8]])
        rows = dk.max(axis=1); print('worst rows', np.argsort(-rows)[:8], rows[np.argsort(-rows)[:8]])
        gb = tr.gs.gradients()[3].cpu().numpy(); rb = want['grads_state'][3]
        print('bias grad diff', np.abs(gb - rb)[:16], 'ref', rb[:8])
        gg = tr.gs.gradients()[0].cpu().numpy(); rg = want['grads_state'][0]
        print('gamma grad diff worst', np.argsort(-np.abs(gg - rg))[:6], np.sort(np.abs(gg - rg))[-6:], 'ref scale', np.abs(rg).max())
