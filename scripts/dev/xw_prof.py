"""Phase totals of k_state_xwide (library built with -DXW_PROFILE, LD_PRELOADed): python scripts/dev/xw_prof.py N E d iterations"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gnnkeras_amd import _native as nat
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
lib = ctypes.CDLL(os.environ['LD_PRELOAD'])
out = (ctypes.c_ulonglong * 8)()
k, st, o = gnn.Loop(*inputs, state0=s0); torch.cuda.synchronize()
lib.gnn_xw_profile(out, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); k, st, o = gnn.Loop(*inputs, state0=s0); e1.record(); torch.cuda.synchronize()
lib.gnn_xw_profile(out, 1)
v = list(out)
wg = 256 * K
tiles = max(v[3], 1)
print(f'k {float(k)}  loop {e0.elapsed_time(e1) / K * 1e3:.0f} us per iteration (instrumented)')
print(f'matrix wave 0, cycles per tile: wait for the slot {v[0] / tiles:.0f}; K loop {v[1] / tiles:.0f}; epilogue {v[2] / tiles:.0f}; tiles per workgroup and launch {tiles / wg:.1f}')
print(f'per launch and workgroup: matrix wave 0 {v[6] / wg:.0f} cycles; gather wave 8 {v[5] / wg:.0f} cycles of which waiting for a free slot {v[4] / wg:.0f}')
