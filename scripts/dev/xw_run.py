"""A short run of the loop at a wide state (profiling target): python scripts/dev/xw_run.py N E d iterations"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gnnkeras_amd.synth import er_graph
from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
from gnnkeras_amd.Models.GNN import GNNnodeBased
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
N, E, d, K = int(float(sys.argv[1])), int(float(sys.argv[2])), int(sys.argv[3]), int(sys.argv[4])
g = er_graph(N, E, aggregation_mode='average')
x = MultiGraphSequencer([g], 'n', 'average', 1, shuffle=False)[0][0]
inp, lay = get_inout_dims('state', 14, 3, 2, 'n', d); ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0)
inp, lay = get_inout_dims('output', 14, 3, 2, 'n', d); no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1)
s0 = torch.from_numpy(np.random.default_rng(1).normal(0, .1, (N, d)).astype(np.float32)).cuda()
gnn = GNNnodeBased(ns, no, d, K, 0.0)
inputs = gnn.process_inputs(x)
for _ in range(2):
    k, st, o = gnn.Loop(*inputs, state0=s0)
torch.cuda.synchronize()
print('k', float(k))
