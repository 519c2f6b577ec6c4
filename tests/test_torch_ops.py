"""torch.ops.gnnkeras.* — the PyTorch-ROCm custom-op boundary (BASELINE.json north_star).

CPU part: the op library loads, registers the six ops with the agreed schemas, and refuses CPU tensors (no CPU path).
GPU part (-m gpu): the ops called directly, the way a foreign torch caller would, against the oracle; TORCH_CHECK-style
errors for wrong dtype / device / shape."""
import numpy as np
import pytest
import torch

from gnnkeras_amd import ops, _native as nat
from gnnkeras_amd.sparse import SparseMatrix
from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
from gnnkeras_amd.Models.GNN import GNNgraphBased, GNNnodeBased
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer

OPS = ('loop_forward', 'aggregate', 'pool', 'converged', 'state_step', 'mlp_forward')


def test_op_library_registers_every_op():
    lib = ops.load()
    for name in OPS:
        op = getattr(lib, name)
        schema = str(op.default._schema)
        assert schema.startswith(f'gnnkeras::{name}('), schema
    s = str(lib.loop_forward.default._schema)
    assert '-> (Tensor k, Tensor state, Tensor out)' in s and 'Tensor?[] adjacency' in s and 'Tensor[] net_state_weights' in s


def test_ops_have_no_cpu_implementation():
    ops.load()
    with pytest.raises(NotImplementedError):
        torch.ops.gnnkeras.converged(torch.zeros(4, 4), None, 0.1)
    csr = dict(rowptr=torch.zeros(3, dtype=torch.int32), src=torch.zeros(0, dtype=torch.int32), w=None, row_scale=None,
               n_dst=2, n_src=2, nnz=0)
    with pytest.raises(NotImplementedError):
        ops.aggregate(csr, torch.zeros(2, 3))


def _nets(focus, d, device=None):
    inp, lay = get_inout_dims('state', 14, 3, 2, focus, d)
    ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0, device=device)
    inp, lay = get_inout_dims('output', 14, 3, 2, focus, d)
    no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1, device=device)
    return ns, no


@pytest.mark.gpu
def test_loop_forward_op_matches_oracle(mutag_graphs):
    """The raw op, called with plain tensors and lists (no model object in between), on the C2 batch."""
    from oracle.harness import oracle_loop, rel_err
    seq = MultiGraphSequencer(mutag_graphs[:32], 'g', 'average', 32, shuffle=False)
    x = seq[0][0]
    d = 32
    ns, no = _nets('g', d)
    model = GNNgraphBased(ns, no, d, 50, 0.0)
    s0 = np.random.default_rng(1).normal(0, 0.1, (x[0].shape[0], d)).astype(np.float32)
    nodes, arcs = x[0], x[1]
    dev = nodes.device
    adj, an, ng = (SparseMatrix.from_triple(t).device_csr(dev) for t in (x[5], x[6], x[7]))
    csr = lambda c: ([c['rowptr'], c['src'], c['w'], c['row_scale']], [c['n_dst'], c['n_src'], c['nnz']])
    spec = lambda n: [n.input_dim, int(n.batch_normalization), len(n.units)] + n.units + [nat.ACTIVATIONS[a] for a in n.activations]
    out_index = torch.arange(nodes.shape[0], dtype=torch.int32, device=dev)
    ops.load()
    k, state, out = torch.ops.gnnkeras.loop_forward(
        nodes, arcs, *csr(adj), *csr(an), *csr(ng), ns.weights, spec(ns), no.weights, spec(no), 1e-3,
        torch.from_numpy(s0).to(dev), out_index, None, None, d, 50, 0.0, 2, 0, [], [], None, [], [], [], [], [])
    torch.cuda.synchronize()
    k64, st64, o64 = oracle_loop(model, x, s0, np.float64)
    assert float(k) == 50.0 == float(k64)
    assert rel_err(state.cpu().numpy(), st64) <= 1e-5 and rel_err(out.cpu().numpy(), o64) <= 1e-5
    # the model's Loop goes through the same op
    k2, st2, o2 = model.Loop(*model.process_inputs(x), state0=torch.from_numpy(s0).to(dev))
    assert torch.equal(st2, state) and torch.equal(o2, out)


@pytest.mark.gpu
def test_piecewise_ops_match_oracle(mutag_graphs):
    from oracle import gnn_oracle as O
    from oracle.harness import oracle_loop, rel_err, _triple
    seq = MultiGraphSequencer(mutag_graphs[:8], 'g', 'average', 8, shuffle=False)
    x = seq[0][0]
    dev = x[0].device
    N, d = x[0].shape[0], 16
    rng = np.random.default_rng(2)
    adj = SparseMatrix.from_triple(x[5]).device_csr(dev)
    X = rng.normal(size=(N, 5)).astype(np.float32)
    got = ops.aggregate(adj, torch.from_numpy(X).to(dev))
    want = O.sparse_dense_matmul_adjoint(*_triple(x[5]), X, np.float64)
    assert rel_err(got.cpu().numpy(), want) <= 1e-6
    ng = SparseMatrix.from_triple(x[7]).device_csr(dev)
    Y = rng.normal(size=(N, 2)).astype(np.float32)
    assert rel_err(ops.pool(ng, torch.from_numpy(Y).to(dev)).cpu().numpy(), O.sparse_dense_matmul_adjoint(*_triple(x[7]), Y, np.float64)) <= 1e-6
    s, so = rng.normal(size=(N, d)).astype(np.float32), rng.normal(size=(N, d)).astype(np.float32)
    for thr in (0.1, 10.0):
        f = ops.converged(torch.from_numpy(s).to(dev), torch.from_numpy(so).to(dev), thr)
        assert bool(f[0]) == bool(O.condition(0, s, so, 1, thr, np.float64))
    ns, no = _nets('g', d)
    one = GNNgraphBased(ns, no, d, 1, 0.0)
    st1 = oracle_loop(one, x, s, np.float64)[1]
    new, moving = ops.state_step(x[0], x[1], adj, SparseMatrix.from_triple(x[6]).device_csr(dev), ns, torch.from_numpy(s).to(dev), d, 0.0)
    assert rel_err(new.cpu().numpy(), st1) <= 1e-5 and int(moving[0]) == 1
    xin = rng.normal(size=(50, no.input_dim)).astype(np.float32)
    y = ops.mlp_forward(no, torch.from_numpy(xin).to(dev))
    assert rel_err(y.cpu().numpy(), O.mlp_apply(*no.spec(), xin, False, np.float64)) <= 1e-5


@pytest.mark.gpu
def test_op_argument_errors_are_runtime_errors(mutag_graphs):
    seq = MultiGraphSequencer(mutag_graphs[:4], 'g', 'average', 4, shuffle=False)
    x = seq[0][0]
    dev = x[0].device
    adj = SparseMatrix.from_triple(x[5]).device_csr(dev)
    N = x[0].shape[0]
    with pytest.raises(RuntimeError, match='float32'):
        ops.aggregate(adj, torch.zeros((N, 3), dtype=torch.float64, device=dev))
    with pytest.raises(RuntimeError, match='n_src'):
        ops.aggregate(adj, torch.zeros((N + 1, 3), device=dev))
    with pytest.raises(RuntimeError, match='contiguous'):
        ops.aggregate(adj, torch.zeros((3, N), device=dev).t())
    bad = dict(adj, rowptr=adj['rowptr'].to(torch.int64))
    with pytest.raises(RuntimeError, match='int32'):
        ops.aggregate(bad, torch.zeros((N, 3), device=dev))
    ns, no = _nets('g', 8)
    model = GNNgraphBased(ns, no, 8, 3, 0.0)
    with pytest.raises((RuntimeError, ValueError), match='state0'):
        model.Loop(*model.process_inputs(x), state0=torch.zeros((N + 1, 8), device=dev))
    wrong = MLP(ns.input_dim + 1, ns.units, 'selu', 'lecun_normal', 'lecun_normal', rng=0)
    with pytest.raises(RuntimeError, match='in_dim'):
        GNNgraphBased(wrong, no, 8, 3, 0.0).Loop(*model.process_inputs(x), state0=torch.zeros((N, 8), device=dev))
