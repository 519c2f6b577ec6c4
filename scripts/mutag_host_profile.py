"""Where the host time of a MUTAG forward goes: cProfile of 136 Loop() calls (device idle between calls is NOT forced)."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import starter_nets
from gnnkeras_amd.load_MUTAG import load_graphs
from gnnkeras_amd.Models.GNN import GNNgraphBased
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
device = torch.device('cuda', 0)
graphs = load_graphs()
seq = MultiGraphSequencer(graphs, 'g', 'average', 32, shuffle=False, device=device)
ns, no = starter_nets(32, device, 'g')
gnn = GNNgraphBased(ns, no, 32, 50, 0.01)
items = [seq[i][0] for i in range(len(seq))]
inputs = [gnn.process_inputs(x) for x in items]
rng = np.random.default_rng(1)
s0s = [torch.from_numpy(rng.normal(0, 0.1, (x[0].shape[0], 32)).astype(np.float32)).to(device) for x in items]
for inp, s0 in zip(inputs, s0s): gnn.Loop(*inp, state0=s0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for rep in range(5):
    for inp, s0 in zip(inputs, s0s): gnn.Loop(*inp, state0=s0)
t_host = (time.perf_counter() - t0) / (5 * len(inputs))
torch.cuda.synchronize()
print(f'host enqueue time per Loop(): {t_host*1e6:.1f} us')
pr = cProfile.Profile(); pr.enable()
for rep in range(5):
    for inp, s0 in zip(inputs, s0s): gnn.Loop(*inp, state0=s0)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
