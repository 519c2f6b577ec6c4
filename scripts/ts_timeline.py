"""Where an iteration of the persistent small-graph training kernels goes: phase times of workgroup 0 (wall_clock64, 10 ns ticks, summed over
the iterations of one launch).  Library built with -DGNN_TS_TIMELINE (`make -C gnnkeras_amd/csrc profile` -> libgnnloop_timeline.so), loaded
through GNNKERAS_AMD_LIB."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gnnkeras_amd import _native as nat
from gnnkeras_amd.load_MUTAG import load_graphs
from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
from gnnkeras_amd.Models.GNN import GNNgraphBased
from gnnkeras_amd.Models.training import Adam
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
d, it, bn = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (32, 50, 1)
gs = load_graphs(limit=32 * 4)
for g in gs: g.setAggregation('average')
seq = MultiGraphSequencer(gs, 'g', 'average', 32, shuffle=False)
inp, lay = get_inout_dims('state', 14, 3, 2, 'g', d); ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0, batch_normalization=bool(bn))
inp, lay = get_inout_dims('output', 14, 3, 2, 'g', d); no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1, batch_normalization=bool(bn))
gnn = GNNgraphBased(ns, no, d, it, 0.01)
gnn.compile(optimizer=Adam(0.01), loss='categorical_crossentropy', metrics=['accuracy'])
L = nat.lib()
L.gnn_ts_phase_times.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 32)()
for rep in range(2):
    for i in range(len(seq)): r = gnn.train_step(seq[i], seed=0)
torch.cuda.synchronize()
assert L.gnn_ts_phase_times(buf) == 0
t = np.array(list(buf), dtype=np.float64).reshape(2, 16) * 0.01 / max(r['k'], 1)        # us per iteration
x = seq[len(seq) - 1][0]
print(f'd = {d}, {r["k"]} iterations, BatchNormalization {bool(bn)}, batch of {x[0].shape[0]} nodes; us per iteration in workgroup 0 (this build):')
fw = ['gather + tape', 'statistics partials', 'barrier 1', 'totals, a / c', 'MFMA + epilogue + stores', 'barrier 2']
bw = ['xhat + dZ (rows prefetched)', 'dy = dZ . W^T (MFMA)', 'S1, S2 of dy, partials', 'Phat (MFMA) + barrier 1', 'totals, coefficients', 'BN gradient + stores', 'barrier 2',
      'gather by source']
print('forward : ' + '; '.join(f'{n} {v:.2f}' for n, v in zip(fw, t[0])) + f'; total {t[0].sum():.2f}')
print('backward: ' + '; '.join(f'{n} {v:.2f}' for n, v in zip(bw, t[1])) + f'; total {t[1].sum():.2f}')
