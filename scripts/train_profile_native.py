"""Kernel-trace target: 20 in-library train steps (gnn_train_step) on MUTAG batches, d = 32, 50 iterations."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gnnkeras_amd.load_MUTAG import load_graphs
from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
from gnnkeras_amd.Models.GNN import GNNgraphBased
from gnnkeras_amd.Models.training import Adam
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
gs = load_graphs(limit=32 * 20)
for g in gs: g.setAggregation('average')
seq = MultiGraphSequencer(gs, 'g', 'average', 32, shuffle=False)
d, it = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 50)
inp, lay = get_inout_dims('state', 14, 3, 2, 'g', d); ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0)
inp, lay = get_inout_dims('output', 14, 3, 2, 'g', d); no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1)
gnn = GNNgraphBased(ns, no, d, it, 0.01)
gnn.compile(optimizer=Adam(0.01), loss='categorical_crossentropy', metrics=['accuracy'])
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(len(seq)): r = gnn.train_step(seq[i], seed=0)
    torch.cuda.synchronize(); print(f'{(time.perf_counter() - t0) / len(seq) * 1e3:.2f} ms/step')
