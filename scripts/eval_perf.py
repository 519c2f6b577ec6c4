"""evaluate() / predict() wall time over the MUTAG data set (136 batches of 32), starter and d = 32 configurations."""
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
for d, K in ((0, 5), (32, 50)):
    inp, lay = get_inout_dims('state', 14, 3, 2, 'g', d); ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0, device=dev)
    inp, lay = get_inout_dims('output', 14, 3, 2, 'g', d); no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1, device=dev)
    gnn = GNNgraphBased(ns, no, d, K, 0.01)
    gnn.compile(optimizer='adam', loss='categorical_crossentropy', metrics=['accuracy'])
    for name, fn in (('predict', lambda: gnn.predict(seq)), ('evaluate', lambda: gnn.evaluate(seq))):
        fn(); fn()
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f'd={d} max_iter={K}: {name:8s} {1e3 * min(ts):7.2f} ms for {len(gs)} graphs = {1e3 * min(ts) / len(gs):.5f} ms/graph')
