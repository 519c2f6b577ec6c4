"""The same with node labels N(DBG_OFF, 1) / N(-0.4 DBG_OFF, 0.5): how far from zero the inputs may sit (DBG_N nodes, DBG_ACT).
(profiles/r04_notes.txt section 7 quotes it.)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from gnnkeras_amd import GraphObject
from gnnkeras_amd.synth import er_graph
from gnnkeras_amd.Models.GNN import GNNnodeBased
from gnnkeras_amd.Models.training import LoopTrainer, SGD
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
from test_gpu_training import nets, oracle_step
OFF = float(os.environ.get('DBG_OFF', '30')); ACT = os.environ.get('DBG_ACT', 'relu')
rng = np.random.default_rng(12)
N = int(os.environ.get('DBG_N', '36000'))
g = er_graph(N, 5 * N, seed=7, aggregation_mode='average')
nodes = g.nodes.copy(); nodes[:, :6] = rng.normal(OFF, 1.0, (N, 6)); nodes[:, 6:] = rng.normal(-0.4 * OFF, 0.5, (N, nodes.shape[1] - 6))
t = np.zeros((N, 2)); t[np.arange(N), rng.integers(0, 2, N)] = 1
g = GraphObject(nodes, g.arcs, t, focus='n', aggregation_mode='average')
x, y, sw = MultiGraphSequencer([g], 'n', 'average', 1, shuffle=False)[0]
ns, no = nets('n', 32, True, act=ACT, scale=0.5)
s0 = np.abs(rng.normal(0, 0.1, (N, 32))).astype(np.float32)
model = GNNnodeBased(ns, no, 32, 3, 0.0)
model.compile(optimizer=SGD(0.0), loss='categorical_crossentropy')
want = oracle_step(model, x, y, sw, s0, 'categorical_crossentropy', False)
for native in (True, False):
    tr = LoopTrainer(model); tr.use_native_step = native
    res = tr.train_step(x, y, sw, state0=torch.from_numpy(s0).cuda(), apply=False)
    errs = []
    for got, ref in [(tr.gs.gradients(), want['grads_state']), (tr.go.gradients(), want['grads_output'])]:
        for g_, r in zip(got, ref):
            errs.append(float(np.max(np.abs(g_.cpu().numpy() - r)) / max(float(np.max(np.abs(r))), 1e-12)))
    print('native' if native else 'blocks', 'off', OFF, ACT, 'k', res['k'], 'loss err', abs(float(res['loss']) - want['loss']), 'y_pred err', float(np.max(np.abs(res['y_pred'].cpu().numpy() - want['y_pred']))), ['%.1e' % e for e in errs])
