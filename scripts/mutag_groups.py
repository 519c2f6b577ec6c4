"""MUTAG forward throughput: grouped launches (merged batches as independent loops) against side streams."""
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
d = int(sys.argv[1]) if len(sys.argv) > 1 else 32          # 0 = the starter configuration (state = the 14 label columns, 5 iterations)
K_IT = 50 if d else 5
inp, lay = get_inout_dims('state', 14, 3, 2, 'g', d); ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0, device=dev)
inp, lay = get_inout_dims('output', 14, 3, 2, 'g', d); no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1, device=dev)
for thr, nm in ((0.01, 'random weights'), (None, 'converging')):
    if thr is None:
        w = ns.get_weights(); ns.set_weights([a * 0.25 if a.ndim == 2 else a for a in w]); thr = 0.01
    gnn = GNNgraphBased(ns, no, d, K_IT, thr)
    for grouped in (False, True):
        gnn.group_batches = grouped
        plan = gnn._group_plan(seq, dev)
        for rep in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            gnn._k_seen = []
            outs = [o for _, o in gnn._forward_batches(seq, dev)]
            ks = torch.cat([k.reshape(-1) for k in gnn._k_seen])
            torch.cuda.synchronize(); t = time.perf_counter() - t0
        print(f'{nm:24s} grouped={grouped!s:5s} launches={len(plan) if plan else len(seq):3d}: {1e6 * t / len(seq):7.1f} us/batch = {1e3 * t / len(gs):.5f} ms/graph, mean k {float(ks.mean()):.1f}')
