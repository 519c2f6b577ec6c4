"""Experiment: phase cycle totals of the wave-specialised fused kernel (library built with -DGNN_F4_PROFILE)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['GNN_FUSED_KERNEL'] = '4'
import numpy as np, torch
from gnnkeras_amd import _native as nat
from gnnkeras_amd.synth import er_graph
from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
from gnnkeras_amd.Models.GNN import GNNnodeBased
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
E, d = 10 * N, 64
g = er_graph(N, E, aggregation_mode='average'); seq = MultiGraphSequencer([g], 'n', 'average', 1, shuffle=False); x = seq[0][0]
inp, lay = get_inout_dims('state', 14, 3, 2, 'n', d); ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0)
inp, lay = get_inout_dims('output', 14, 3, 2, 'n', d); no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1)
s0 = torch.from_numpy(np.random.default_rng(1).normal(0, .1, (N, d)).astype(np.float32)).cuda()
gnn = GNNnodeBased(ns, no, d, 20, 0.0); inputs = gnn.process_inputs(x)
gnn.Loop(*inputs, state0=s0); torch.cuda.synchronize()
L = nat.lib(); buf = (ctypes.c_ulonglong * 8)()
L.gnn_f4_profile(buf, 1)
gnn.Loop(*inputs, state0=s0); torch.cuda.synchronize()
L.gnn_f4_profile(buf, 0)
v = list(buf)
jobs, tiles = 20 * (N // 4), 20 * (N // 16)
n_wg = 512
unit = (v[6] / (n_wg * 12 * 20))            # counter units per launch (gather wave lifetime)
print('units per launch (gather wave lifetime): %.0f ; matrix wave lifetime %.0f' % (unit, v[7] / (n_wg * 4 * 20)))
print('per gather job : gather %.1f%%, slot wait %.1f%% of lifetime' % (100 * v[0] / v[6], 100 * v[1] / v[6]))
print('per matrix tile: fill wait %.1f%%, C+mfma %.1f%%, epilogue %.1f%% of lifetime; per tile units: wait %.0f mfma %.0f epi %.0f' % (
    100 * v[2] / v[7], 100 * v[3] / v[7], 100 * v[4] / v[7], v[2] / tiles, v[3] / tiles, v[4] / tiles))
