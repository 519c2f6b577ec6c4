"""Batch assembly on the device (gnnkeras_amd/device_batch.py: the data set uploaded once, a merged batch = one ragged-copy
launch) against the host path (numpy `GraphObject.merge` as in the reference, graph_class.py:386-413, pinned by the golden
fixtures): the same arrays, the same by-destination CSRs, the same results through the model and through a training step."""
import numpy as np
import pytest
import torch

from gnnkeras_amd import GraphObject
from gnnkeras_amd.device_batch import DeviceDataset
from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
from gnnkeras_amd.Models.GNN import GNNnodeBased, GNNarcBased, GNNgraphBased
from gnnkeras_amd.Models.training import Adam, SGD, LoopTrainer
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer, CompositeMultiGraphSequencer
from gnnkeras_amd.sparse import SparseMatrix

pytestmark = pytest.mark.gpu
CLS = {'n': GNNnodeBased, 'a': GNNarcBased, 'g': GNNgraphBased}


def refocus(graphs, focus, rng):
    if focus == 'g': return [g.copy() for g in graphs]
    out = []
    for g in graphs:
        n = (g.nodes if focus == 'n' else g.arcs).shape[0]
        om = rng.random(n) < 0.7
        t = np.zeros((int(om.sum()), 2)); t[np.arange(len(t)), rng.integers(0, 2, len(t))] = 1
        out.append(GraphObject(nodes=g.nodes, arcs=g.arcs, targets=t, focus=focus, set_mask=rng.random(n) < 0.8, output_mask=om,
                               sample_weight=rng.uniform(0.5, 1.5, len(t))))
    return out


def eff_scale(c, n_dst):
    deg = np.diff(c['rowptr'].cpu().numpy())
    if c['w'] is not None: return c['w'].cpu().numpy()
    s = np.ones(n_dst, np.float32) if c['row_scale'] is None else c['row_scale'].cpu().numpy()
    return np.repeat(s, deg)                  # per entry; rows without entries never matter


@pytest.mark.parametrize('focus', ['g', 'n', 'a'])
@pytest.mark.parametrize('mode', ['average', 'sum', 'normalized'])
def test_device_assembled_batches_equal_host_merged_ones(mutag_graphs, focus, mode):
    rng = np.random.default_rng(3)
    gl = refocus(mutag_graphs[:150], focus, rng)
    for g in gl: g.setAggregation(mode)
    host = MultiGraphSequencer(gl, focus, mode, 32, shuffle=False, assemble='host')
    dev = MultiGraphSequencer(gl, focus, mode, 32, shuffle=False, assemble='device')
    assert len(host) == len(dev) == 5 and type(dev.graph_tensors[0]).__name__ == 'DeviceBatch'
    for i in range(len(host)):
        (xh, yh, wh), (xd, yd, wd) = host[i], dev[i]
        for a, b in zip(xh[:5], xd[:5]):
            assert a.shape == b.shape and a.dtype == b.dtype and torch.equal(a.cpu(), b.cpu())
        assert torch.equal(yh, yd) and torch.equal(wh, wd)
        from gnnkeras_amd.device_batch import lookup_out_index
        want = torch.nonzero(torch.logical_and(xd[3].squeeze(-1), xd[4].squeeze(-1))).reshape(-1).to(torch.int32)
        assert torch.equal(lookup_out_index(xd[3].squeeze(-1), xd[4].squeeze(-1)), want)
        for j in (5, 6) + ((7,) if focus == 'g' else ()):
            mh, md = SparseMatrix.from_triple(xh[j]), SparseMatrix.from_triple(xd[j])
            ch, cd = mh.device_csr('cuda'), md.device_csr('cuda')
            assert (ch['n_dst'], ch['n_src'], ch['nnz']) == (cd['n_dst'], cd['n_src'], cd['nnz'])
            assert torch.equal(ch['rowptr'], cd['rowptr']) and torch.equal(ch['src'], cd['src'])
            assert np.array_equal(eff_scale(ch, ch['n_dst']), eff_scale(cd, cd['n_dst']))
            # the COO triple of the reference's sequencer tuple, rebuilt lazily from the device CSR
            assert np.array_equal(mh.indices, md.indices) and np.array_equal(mh.values, md.values) and mh.shape == md.shape
            ih, vh, sh = xh[j]; idd, vd, sd = xd[j]
            assert torch.equal(ih.cpu(), idd.cpu()) and torch.equal(vh.cpu(), vd.cpu()) and torch.equal(sh, sd)


@pytest.mark.parametrize('focus', ['g', 'n', 'a'])
def test_model_and_training_step_agree_on_both_assemblies(mutag_graphs, focus):
    rng = np.random.default_rng(5)
    gl = refocus(mutag_graphs[:64], focus, rng)
    for g in gl: g.setAggregation('average')
    host = MultiGraphSequencer(gl, focus, 'average', 32, shuffle=False, assemble='host')
    dev = MultiGraphSequencer(gl, focus, 'average', 32, shuffle=False, assemble='device')
    d = 16
    inp, lay = get_inout_dims('state', 14, 3, 2, focus, d); ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0)
    inp, lay = get_inout_dims('output', 14, 3, 2, focus, d); no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1)
    model = CLS[focus](ns, no, d, 8, 0.0)
    model.compile(optimizer=SGD(0.0), loss='categorical_crossentropy')
    for i in range(2):
        s0 = torch.randn(host[i][0][0].shape[0], d, device='cuda') * 0.1
        kh, sth, oh = model.Loop(*model.process_inputs(host[i][0]), state0=s0)
        kd, std, od = model.Loop(*model.process_inputs(dev[i][0]), state0=s0)
        assert float(kh) == float(kd) and torch.equal(sth, std) and torch.equal(oh, od)
        for native in (True, False):          # both orchestrations of the training step read the by-source operands
            grads = []
            for seq in (host, dev):
                tr = LoopTrainer(model); tr.use_native_step = native
                w0 = [w.copy() for w in ns.get_weights() + no.get_weights()]
                res = tr.train_step(*seq[i], state0=s0, apply=False)
                grads.append([g.clone() for g in tr.gs.gradients() + tr.go.gradients()] + [res['loss'].clone()])
                ns.set_weights(w0[:len(ns.get_weights())]); no.set_weights(w0[len(ns.get_weights()):])
            for a, b in zip(*grads):
                assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)       # scatter-add of arc end points uses float atomics


def test_epoch_reshuffle_rebuilds_on_device_and_fit_runs(mutag_graphs):
    gl = [g.copy() for g in mutag_graphs[:200]]
    for g in gl: g.setAggregation('average')
    seq = MultiGraphSequencer(gl, 'g', 'average', 32, shuffle=True)           # 'auto' = device on a GPU box
    assert type(seq.graph_tensors[0]).__name__ == 'DeviceBatch'
    first = seq[0][0][0].clone()
    np.random.seed(4)
    seq.on_epoch_end()
    assert type(seq.graph_tensors[0]).__name__ == 'DeviceBatch' and not torch.equal(first[:50], seq[0][0][0][:50])
    # every graph still appears exactly once per epoch: compare against a host merge of the reshuffled list
    ref = MultiGraphSequencer(list(seq.data), 'g', 'average', 32, shuffle=False, assemble='host')
    for i in range(len(seq)):
        assert torch.equal(seq[i][0][0], ref[i][0][0]) and torch.equal(seq[i][0][1], ref[i][0][1]) and torch.equal(seq[i][1], ref[i][1])
    inp, lay = get_inout_dims('state', 14, 3, 2, 'g', 0); ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0)
    inp, lay = get_inout_dims('output', 14, 3, 2, 'g', 0); no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1)
    model = GNNgraphBased(ns, no, 0, 5, 0.01)
    model.compile(optimizer=Adam(0.01), loss='categorical_crossentropy', metrics=['accuracy'])
    hist = model.fit(seq, epochs=4, verbose=0)
    assert hist['loss'][-1] < hist['loss'][0]
    assert seq.copy().assemble == 'auto' and isinstance(CompositeMultiGraphSequencer, type)


def test_device_assembly_refuses_what_it_does_not_cover(mutag_graphs):
    with pytest.raises(ValueError):
        DeviceDataset(mutag_graphs[:4], 'g', 'composite_average', 'cuda')
    g = mutag_graphs[0]
    odd = GraphObject(nodes=np.ones((3, 5)), arcs=np.array([[0, 1, 1.], [1, 2, 1.]]), targets=np.ones((1, 2)), focus='g')
    seq = MultiGraphSequencer([g, odd], 'g', 'sum', 2, shuffle=False) if False else None      # different widths cannot merge at all
    with pytest.raises(ValueError):
        DeviceDataset([g, odd], 'g', 'sum', 'cuda')
