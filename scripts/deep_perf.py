"""us per iteration of a state network with hidden layers on an ER graph: the first two Dense layers fused with the aggregate against the
un-fused path.   python scripts/deep_perf.py [N] [E] [d] [h1,h2,..]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gnnkeras_amd import _native as nat
from gnnkeras_amd.synth import er_graph
from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
from gnnkeras_amd.Models.GNN import GNNnodeBased
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
E = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10 * N
d = int(sys.argv[3]) if len(sys.argv) > 3 else 64
hidden = [int(v) for v in sys.argv[4].split(',')] if len(sys.argv) > 4 else [64, 64]
K = 20
g = er_graph(N, E, aggregation_mode='average'); seq = MultiGraphSequencer([g], 'n', 'average', 1, shuffle=False); x = seq[0][0]
inp, lay = get_inout_dims('state', 14, 3, 2, 'n', d, hidden_units=hidden); ns = MLP(inp[0], lay, 'tanh', 'lecun_normal', 'lecun_normal', rng=0)
inp, lay = get_inout_dims('output', 14, 3, 2, 'n', d); no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1)
s0 = torch.from_numpy(np.random.default_rng(1).normal(0, .1, (N, d)).astype(np.float32)).cuda()
gnn = GNNnodeBased(ns, no, d, K, 0.0); inputs = gnn.process_inputs(x)
for name, flags in (('fused prefix', 0), ('un-fused', nat.FLAG_UNFUSED)):
    gnn.native_flags = flags
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    for e in ev: e.record()
    gnn.loop_events = ev
    for rep in range(3):
        k, st, o = gnn.Loop(*inputs, state0=s0); torch.cuda.synchronize()
    us = 1e3 * ev[0].elapsed_time(ev[1]) / K
    print(f'N={N} E={E} d={d} hidden={hidden} {name:13s}: {us:8.1f} us per iteration  ({nat.lib().gnn_last_kernel_name().decode()})')
