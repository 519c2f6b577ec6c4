"""Host-side profile of a fit() epoch on MUTAG in the reference's starter configuration (state_vect_dim = 0, 5 iterations): where the
Python time of an epoch goes, next to the device time of its kernels (scripts/fit_perf.py gives the wall time)."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gnnkeras_amd.load_MUTAG import load_graphs
from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
from gnnkeras_amd.Models.GNN import GNNgraphBased
from gnnkeras_amd.Models.training import Adam
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
d, it = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 5)
bs = int(sys.argv[3]) if len(sys.argv) > 3 else 32            # (the reference's starter.py default is 1000)
gs = load_graphs()
for g in gs: g.setAggregation('average')
tr = MultiGraphSequencer(gs[:-868], 'g', 'average', bs, shuffle=True, device='cuda')
va = MultiGraphSequencer(gs[-868:], 'g', 'average', bs, shuffle=False, device='cuda')
inp, lay = get_inout_dims('state', 14, 3, 2, 'g', d); ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0)
inp, lay = get_inout_dims('output', 14, 3, 2, 'g', d); no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1)
gnn = GNNgraphBased(ns, no, d, it, 0.01)
gnn.compile(optimizer=Adam(0.01), loss='categorical_crossentropy', metrics=['accuracy'])
gnn.fit(tr, epochs=1, validation_data=va, verbose=0)
torch.cuda.synchronize(); t0 = time.perf_counter()
gnn.fit(tr, epochs=3, validation_data=va, verbose=0)
torch.cuda.synchronize(); print(f'd={d} it={it} batch_size={bs}: fit epoch {1e3 * (time.perf_counter() - t0) / 3:.1f} ms ({len(tr)} steps)')
# steps alone, host launch time vs completion
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(len(tr)): gnn.train_step(tr[i])
t_host = time.perf_counter() - t0
torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print(f'{len(tr)} train steps: host returned after {1e3 * t_host:.1f} ms, device done after {1e3 * t_all:.1f} ms ({1e3 * t_all / len(tr):.3f} ms per step)')
t0 = time.perf_counter(); tr.on_epoch_end(); torch.cuda.synchronize(); print(f'on_epoch_end {1e3 * (time.perf_counter() - t0):.1f} ms')
t0 = time.perf_counter(); gnn.evaluate(va); torch.cuda.synchronize(); print(f'evaluate(validation) {1e3 * (time.perf_counter() - t0):.1f} ms')
pr = cProfile.Profile(); pr.enable()
gnn.fit(tr, epochs=1, validation_data=va, verbose=0)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(25)
