import sys, os, cProfile, pstats
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from gnnkeras_amd.load_MUTAG import load_graphs
from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
from gnnkeras_amd.Models.GNN import GNNgraphBased
from gnnkeras_amd.Models.training import Adam
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
gs = load_graphs(limit=32 * 8)
for g in gs: g.setAggregation('average')
seq = MultiGraphSequencer(gs, 'g', 'average', 32, shuffle=False)
d, it = 32, 50
inp, lay = get_inout_dims('state', 14, 3, 2, 'g', d); ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0)
inp, lay = get_inout_dims('output', 14, 3, 2, 'g', d); no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1)
gnn = GNNgraphBased(ns, no, d, it, 0.01)
gnn.compile(optimizer=Adam(0.01), loss='categorical_crossentropy', metrics=['accuracy'])
for i in range(len(seq)): gnn.train_step(seq[i], seed=0)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(len(seq)): gnn.train_step(seq[i], seed=0)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
