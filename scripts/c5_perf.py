"""C5 probe: composite (3 node types) ER 500k nodes / 5M arcs, d=64, per-type state nets; per-iteration time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gnnkeras_amd.synth import er_composite_graph
from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
from gnnkeras_amd.Models.CompositeGNN import CompositeGNNnodeBased
from gnnkeras_amd.Sequencers.GraphSequencers import CompositeMultiGraphSequencer
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 500000
E = int(float(sys.argv[2])) if len(sys.argv) > 2 else 5000000
mode = sys.argv[3] if len(sys.argv) > 3 else 'average'
d, dims = 64, (14, 8, 4)
t = time.time(); g = er_composite_graph(N, E, dim_node_label=dims, aggregation_mode=mode); print('graph', round(time.time() - t, 1), 's')
seq = CompositeMultiGraphSequencer([g], 'n', mode, 1, shuffle=False); x = seq[0][0]
inp, lay = get_inout_dims('state', dims, 3, 2, 'n', d)
ns = [MLP(i, lay, 'selu', 'lecun_normal', 'lecun_normal', rng=t_) for t_, i in enumerate(inp)]
inp, lay = get_inout_dims('output', dims, 3, 2, 'n', d)
no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=9)
s0 = torch.from_numpy(np.random.default_rng(1).normal(0, .1, (N, d)).astype(np.float32)).cuda()
def run(iters, reps=3):
    gnn = CompositeGNNnodeBased(ns, no, d, iters, 0.0)
    inputs = gnn.process_inputs(x); ts = []
    for r in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); k, st, o = gnn.Loop(*inputs, state0=s0); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts[1:]), float(k)
per_arc = x[7].matrix.csr().w is not None
B = E * (4 + 4 * d + (4 if per_arc else 0)) + N * (4 + 4 + 12 * d)      # + 4 B type-list entry per node
t10, _ = run(10); t50, k = run(50)
it = (t50 - t10) / 40
print(f'C5 {mode}: per_arc_w={per_arc} fwd(50)={t50:.2f} ms k={k} -> {it*1e3:.1f} us/iter ({it*1e3/3:.1f} us per type launch), '
      f'{E/it/1e6:.2f} G arc-updates/s, algorithmic {B/it/1e6:.0f} GB/s = {B/it/8e7:.1f}% of 8 TB/s')
