"""Where does the MUTAG predict() walk spend its time? Each plan entry alone (events around it), then all together."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
plan = gnn._group_plan(seq, dev)
print('plan:', [(len(bs), bs.resident, sum((seq[b][0][0].shape[0] + 63) // 64 for b in bs)) for bs in plan])
for li, bs in enumerate(plan):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = gnn._plan_launch(seq, bs); e1.record()
        th = time.perf_counter() - t0
        torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f'entry {li}: {len(bs)} batches resident={bs.resident}: host {1e3 * th:.3f} ms, wall {1e3 * t:.3f} ms, device {e0.elapsed_time(e1):.3f} ms')
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gnn._k_seen = []
    outs = [o for _, o in gnn._forward_batches(seq, dev)]
    th = time.perf_counter() - t0
    torch.cuda.synchronize(); t = time.perf_counter() - t0
print(f'whole walk: host {1e3 * th:.3f} ms, wall {1e3 * t:.3f} ms')
