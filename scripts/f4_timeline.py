"""Evidence for the C3 (100 k / 1 M) gap: per-workgroup s_memtime stamps of ONE launch of the wave-specialised kernel (library built
with -DGNN_F4_TIMELINE: `make -C gnnkeras_amd/csrc profile` -> libgnnloop_timeline.so, loaded through GNNKERAS_AMD_LIB): when every workgroup entered, finished
its W1 fill, made its first deposit and left (wall_clock64: 10 ns resolution)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['GNN_FUSED_KERNEL'] = '4'
import numpy as np, torch
from gnnkeras_amd import _native as nat
from gnnkeras_amd.synth import er_graph
from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
from gnnkeras_amd.Models.GNN import GNNnodeBased
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000
E, d = 10 * N, 64
g = er_graph(N, E, aggregation_mode='average'); seq = MultiGraphSequencer([g], 'n', 'average', 1, shuffle=False); x = seq[0][0]
inp, lay = get_inout_dims('state', 14, 3, 2, 'n', d); ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0)
inp, lay = get_inout_dims('output', 14, 3, 2, 'n', d); no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1)
s0 = torch.from_numpy(np.random.default_rng(1).normal(0, .1, (N, d)).astype(np.float32)).cuda()
gnn = GNNnodeBased(ns, no, d, 6, 0.0); inputs = gnn.process_inputs(x)
L = nat.lib()
L.gnn_f4_wg_times.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 4096)()
ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
for e in ev: e.record()
gnn.loop_events = ev
for rep in range(3):
    gnn.Loop(*inputs, state0=s0); torch.cuda.synchronize()
launch_us = 1e3 * ev[0].elapsed_time(ev[1]) / 6          # this (instrumented) build, per launch
assert L.gnn_f4_wg_times(buf) == 0
t = np.array(list(buf), dtype=np.float64).reshape(1024, 4)
t = t[t[:, 3] > 0]
n_wg = len(t)
# wall_clock64() = s_memrealtime: 100 MHz, one base for the whole device
rel = t - t[:, 0].min()
tick_us = 0.01
ent, fil, dep, ext = (rel[:, i] * tick_us for i in range(4))
q = lambda a: 'min %6.1f  p10 %6.1f  p50 %6.1f  p90 %6.1f  max %6.1f' % (a.min(), np.percentile(a, 10), np.percentile(a, 50), np.percentile(a, 90), a.max())
print(f'N = {N}: {n_wg} workgroups, one launch = {launch_us:.1f} us in this build (four time stamps per workgroup); '
      f'us after the first workgroup entered')
print('entry         :', q(ent))
print('W1 fill done  :', q(fil), ' (median %.1f us after entry)' % np.median(fil - ent))
print('first deposit :', q(dep), ' (median %.1f us after the fill)' % np.median(dep - fil))
print('exit          :', q(ext))
print('lifetime      :', q(ext - ent))
busy = np.array([((ent <= x_) & (ext > x_)).sum() for x_ in np.linspace(0, ext.max(), 101)])
print('workgroups alive at 0 %% .. 100 %% of the launch, every 10 %%: %s' % ' '.join(str(int(v)) for v in busy[::10]))
print(f'all {n_wg} workgroups busy from {ent.max():.1f} to {ext.min():.1f} us = {100 * max(ext.min() - ent.max(), 0) / ext.max():.0f} % of the launch; '
      f'ramp-up (last entry + fill + first deposit) {np.percentile(dep, 90):.1f} us; tail (max exit - median exit) {ext.max() - np.median(ext):.1f} us')
if os.environ.get('F4_RAW'):
    life = t[:, 3] - t[:, 0]
    print('raw lifetime ticks:', q(life))
    order = np.argsort(t[:, 0])
    gaps = np.diff(t[order, 0])
    print('largest gaps between sorted entry stamps:', np.sort(gaps)[-10:])
    for b in range(0, 24): print(b, [int(v) for v in t[b]])
xcd = np.arange(n_wg) % 8
print('exit per XCD (min / median / max us): ' + '  '.join('%d: %.0f/%.0f/%.0f' % (x_, ext[xcd == x_].min(), np.median(ext[xcd == x_]), ext[xcd == x_].max()) for x_ in range(8)))
