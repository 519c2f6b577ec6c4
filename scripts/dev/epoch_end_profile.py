"""cProfile of MultiGraphSequencer.on_epoch_end() (reshuffle + device re-merge of all 136 MUTAG batches)."""
import sys, os, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gnnkeras_amd.load_MUTAG import load_graphs
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
gs = load_graphs()
seq = MultiGraphSequencer(gs[:-868], 'g', 'average', 32, shuffle=True)
for _ in range(3): seq.on_epoch_end()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): seq.on_epoch_end()
torch.cuda.synchronize(); print(f'on_epoch_end: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms')
pr = cProfile.Profile(); pr.enable()
for _ in range(10): seq.on_epoch_end()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
import gc
gc.disable()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): seq.on_epoch_end()
torch.cuda.synchronize(); print(f'on_epoch_end with the cyclic GC paused: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms')
gc.enable()
