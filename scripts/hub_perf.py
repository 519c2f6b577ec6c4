"""Skew probe: ER graph plus a few hub nodes; per-iteration time with and without the hub segment pre-pass."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gnnkeras_amd import GraphObject, sparse
from gnnkeras_amd.synth import er_arcs
from gnnkeras_amd.Models.GNN import GNNnodeBased
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
from bench import starter_nets
N, E, d = 1_000_000, 10_000_000, 64
rng = np.random.default_rng(0)
ids = er_arcs(N, E, seed=3)
hubs = [(7, 200_000), (500_000, 50_000), (999_999, 10_000)]
extra = [np.stack([rng.choice(N, deg, replace=False), np.full(deg, h)], 1) for h, deg in hubs]
ids = np.unique(np.concatenate([ids] + extra), axis=0); ids = ids[ids[:, 0] != ids[:, 1]]
arcs = np.concatenate([ids.astype(np.float64), np.eye(3)[rng.integers(0, 3, len(ids))]], axis=1)
g = GraphObject(np.eye(14)[rng.integers(0, 14, N)], arcs, np.zeros((N, 2)), focus='n', aggregation_mode='average')
ns, no = starter_nets(d, 'cuda')
s0 = torch.from_numpy(rng.normal(0, .1, (N, d)).astype(np.float32)).cuda()
for thr in (sparse.HEAVY_THRESHOLD, 10 ** 9):
    sparse.HEAVY_THRESHOLD = thr
    seq = MultiGraphSequencer([g], 'n', 'average', 1, shuffle=False); x = seq[0][0]
    ts = {}
    for iters in (5, 25):
        gnn = GNNnodeBased(ns, no, d, iters, 0.0); inputs = gnn.process_inputs(x)
        best = 1e9
        for r in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gnn.Loop(*inputs, state0=s0); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
        ts[iters] = best
    print(f'hub threshold {thr}: {(ts[25] - ts[5]) / 20 * 1e3:.0f} us / iteration ({len(ids)} arcs, max in-degree {hubs[0][1]})')
