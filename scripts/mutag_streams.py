"""MUTAG forward throughput against the number of side streams predict() / evaluate() use."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gnnkeras_amd.load_MUTAG import load_graphs
from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
from gnnkeras_amd.Models.GNN import GNNgraphBased
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
dev = torch.device('cuda', 0)
gs = load_graphs()
seq = MultiGraphSequencer(gs, 'g', 'average', 32, shuffle=False, device=dev)
d = 32
inp, lay = get_inout_dims('state', 14, 3, 2, 'g', d); ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0, device=dev)
inp, lay = get_inout_dims('output', 14, 3, 2, 'g', d); no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1, device=dev)
gnn = GNNgraphBased(ns, no, d, 50, 0.01)
items = [seq[i][0] for i in range(len(seq))]
inputs = [gnn.process_inputs(x) for x in items]
s0s = [torch.randn((x[0].shape[0], d), device=dev) * 0.1 for x in items]
run = lambda i: gnn.Loop(*inputs[i], state0=s0s[i])
for w in (1, 2, 4, 8, 12, 16):
    for _ in gnn._batches_concurrently(len(inputs), run, dev, w): pass
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in gnn._batches_concurrently(len(inputs), run, dev, w): pass
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f'{w:2d} streams: host enqueue {1e6 * t_host / len(inputs):6.1f} us/batch, total {1e6 * t / len(inputs):6.1f} us/batch = {1e3 * t / len(gs):.4f} ms/graph')
