"""Per-iteration cost of the resident MUTAG launch: device time at two iteration counts; group sizes; kernel name."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gnnkeras_amd import _native as nat
from gnnkeras_amd.load_MUTAG import load_graphs
from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
from gnnkeras_amd.Models.GNN import GNNgraphBased
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
dev = torch.device('cuda', 0)
gs = load_graphs()
seq = MultiGraphSequencer(gs, 'g', 'average', 32, shuffle=False, device=dev)
d = int(sys.argv[1]) if len(sys.argv) > 1 else 32
inp, lay = get_inout_dims('state', 14, 3, 2, 'g', d); ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0, device=dev)
inp, lay = get_inout_dims('output', 14, 3, 2, 'g', d); no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1, device=dev)
res = {}
for K in (10, 50):
    gnn = GNNgraphBased(ns, no, d, K, 0.0)
    plan = gnn._group_plan(seq, dev)
    bs = plan[0]
    sizes = {b: seq[b][0][0].shape[0] for b in bs}
    begin, sets = bs.groups_and_sets(sizes)
    gsz = np.diff(begin)
    ts = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); e1.record(); gnn.loop_events = (e0, e1)
        gnn._plan_launch(seq, bs); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    res[K] = min(ts)
    print(f'K={K}: launches {len(plan)}, groups {len(gsz)} (max {gsz.max()}, mean {gsz.mean():.0f}), sets {None if sets is None else len(sets) - 1}, '
          f'loop {res[K]:.3f} ms, kernel {nat.lib().gnn_last_kernel_name().decode()}')
print(f'per iteration: {(res[50] - res[10]) / 40 * 1e3:.2f} us')
