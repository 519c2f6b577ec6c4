"""cProfile of one serial LGNN fit() epoch in the reference's default configuration (bench.py::lgnn_starter_section): where the host time goes."""
import os, sys, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from gnnkeras_amd.load_MUTAG import load_graphs
from gnnkeras_amd.Models.GNN import GNNgraphBased
from gnnkeras_amd.Models.LGNN import LGNN
from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
from gnnkeras_amd.Models.training import Adam
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
device = torch.device('cuda:0')
graphs = load_graphs()
gnns = []
for i in range(3):
    inp, lay = get_inout_dims('state', 14, 3, 2, 'g', 0, layer=i, get_state=True, get_output=True)
    ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=10 + i, device=device)
    inp, lay = get_inout_dims('output', 14, 3, 2, 'g', 0, layer=i, get_state=True, get_output=True)
    no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=20 + i, device=device)
    gnns.append(GNNgraphBased(ns, no, 0, 5, 0.01))
lg = LGNN(gnns, True, True)
lg.compile(optimizer=Adam(0.01), loss='categorical_crossentropy', average_st_grads=True, metrics=['accuracy'], training_mode='serial')
gs = [g.copy() for g in graphs]
for g in gs: g.setAggregation('average')
tr = MultiGraphSequencer(gs[:-1500], 'g', 'average', 1000, shuffle=True, device=device)
va = MultiGraphSequencer(gs[-750:], 'g', 'average', 1000, shuffle=False, device=device)
lg.fit(tr, epochs=1, validation_data=va, verbose=0)
torch.cuda.synchronize(); t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
lg.fit(tr, epochs=1, validation_data=va, verbose=0)
torch.cuda.synchronize(); pr.disable()
print(f'fit: {time.perf_counter() - t0:.2f} s')
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
