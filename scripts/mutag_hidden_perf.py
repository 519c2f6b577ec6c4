import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from gnnkeras_amd import _native as nat
from gnnkeras_amd.load_MUTAG import load_graphs
from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
from gnnkeras_amd.Models.GNN import GNNgraphBased
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
gs = load_graphs(limit=32 * 8)
seq = MultiGraphSequencer(gs, 'g', 'average', 32, shuffle=False)
d = 32
inp, lay = get_inout_dims('state', 14, 3, 2, 'g', d, hidden_units=[32]); ns = MLP(inp[0], lay, 'tanh', 'lecun_normal', 'lecun_normal', rng=0)
inp, lay = get_inout_dims('output', 14, 3, 2, 'g', d); no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1)
for flags, name in ((0, 'default (persistent, two layers)'), (nat.FLAG_UNFUSED, 'un-fused')):
    gnn = GNNgraphBased(ns, no, d, 50, 0.0); gnn.native_flags = flags
    x = seq[0][0]; inputs = gnn.process_inputs(x)
    s0 = torch.randn((x[0].shape[0], d), device='cuda') * 0.1
    for _ in range(3): gnn.Loop(*inputs, state0=s0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): k, st, o = gnn.Loop(*inputs, state0=s0)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 20
    print(f'{name}: forward {t * 1e6:.0f} us ({t * 1e6 / 50:.1f} us / iteration incl. setup), k={float(k)}')
