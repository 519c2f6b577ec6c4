"""ER roofline probe: times the loop at two iteration counts and reports per-iteration time / algorithmic GB/s."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gnnkeras_amd import _native as nat, GraphTensor
from gnnkeras_amd.synth import er_graph
from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
from gnnkeras_amd.Models.GNN import GNNnodeBased
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000
E = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1000000
d = int(sys.argv[3]) if len(sys.argv) > 3 else 64
mode = sys.argv[4] if len(sys.argv) > 4 else 'average'
hidden = [int(sys.argv[5])] if len(sys.argv) > 5 else None          # optional hidden layer of the state network
t = time.time(); g = er_graph(N, E, aggregation_mode=mode); print('graph build', round(time.time() - t, 1), 's', g)
t = time.time(); seq = MultiGraphSequencer([g], 'n', mode, 1, shuffle=False); x = seq[0][0]; print('sequencer', round(time.time() - t, 1), 's')
inp, lay = get_inout_dims('state', 14, 3, 2, 'n', d, hidden_units=hidden)
ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0)
inp, lay = get_inout_dims('output', 14, 3, 2, 'n', d)
no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1)
s0 = torch.from_numpy(np.random.default_rng(1).normal(0, .1, (N, d)).astype(np.float32)).cuda()
def run(iters, flags=0, reps=3):
    gnn = GNNnodeBased(ns, no, d, iters, 0.0); gnn.native_flags = flags
    inputs = gnn.process_inputs(x)
    ts = []
    for r in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); k, st, o = gnn.Loop(*inputs, state0=s0); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts[1:]), float(k), st
B_iter = E * (4 + 4 * d) + N * (4 + 12 * d)
for flags, nm in [(0, 'fused (GNN_FUSED_KERNEL=%s)' % os.environ.get('GNN_FUSED_KERNEL', 'auto'))] + ([(nat.FLAG_UNFUSED, 'un-fused')] if hidden else []):
    t10, k10, _ = run(10, flags); t50, k50, st = run(50, flags)
    it = (t50 - t10) / 40
    print(f'{nm}: fwd(10)={t10:.2f} ms fwd(50)={t50:.2f} ms k={k50} -> {it*1e3:.1f} us/iter, {E/it/1e6:.2f} G edge-updates/s, '
          f'algorithmic {B_iter/it/1e9*1e3:.0f} GB/s = {B_iter/it/8e9*1e3*100/1e3:.1f}% of 8 TB/s; state absmax {float(st.abs().max()):.3f}')
