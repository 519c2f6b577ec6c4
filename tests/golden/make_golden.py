#!/opt/conda/bin/python3.9
"""Generate the golden fixtures of the graph data layer by RUNNING THE REFERENCE'S OWN numpy/scipy code.

Run in the build container only (the reference never travels to the GPU box):

    /opt/conda/bin/python3.9 tests/golden/make_golden.py

How the reference is made importable: every reference module does `import tensorflow as tf` at import
time, but the numpy/scipy part of `GraphObject` / `CompositeGraphObject` (ctor, buildArcNode, buildAdjacency,
buildNodeGraph, buildCompositeAdjacency, merge, setAggregation) uses TensorFlow for exactly one thing: the dtype string
`tf.keras.backend.floatx()`. A throw-away module named `tensorflow` exposing only that (and an empty `Tensor` class for
one annotation) is created in a temp dir at run time; it never enters this repository. scipy 1.7.1 of the conda
interpreter is required; for the transductive sequencer's static numpy method the stub also carries an
empty `keras.utils.Sequence` base class. Nothing in TensorFlow's arithmetic is exercised or pinned by these fixtures:
they pin rows a13-a16 and a18 of SURVEY.md §8 (graph operands), not the Loop.

The reference's `load_MUTAG.py` cannot run at HEAD: its text is read from /root/reference at run time,
the multi-character delimiter is patched to ',' and the broken composite tail is cut; the rest executes unmodified.

On-disk formats: the reference's own `GraphObject.save` / `savetxt` / `save_dataset` /
`save_dataset_txt` and `CompositeGraphObject.save` (graph_class.py:200-290, composite_graph_class.py:133-138) WRITE a few
small graphs into tests/golden/ref_files/; the tests read those files back with this repository's loaders. The files are
data written by the reference's code, not its source. (`GraphTensor.save_graph` needs TensorFlow ops and cannot run here.)

Outputs (committed): tests/golden/graph_fixtures.npz, tests/golden/ref_files/
"""
import os
import sys
import tempfile
import types

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def _install_tf_stub():
    d = tempfile.mkdtemp(prefix='tfstub_')
    os.makedirs(os.path.join(d, 'tensorflow'))
    with open(os.path.join(d, 'tensorflow', '__init__.py'), 'w') as f:
        f.write("class Tensor: pass\n"
                "class _B:\n"
                "    @staticmethod\n"
                "    def floatx(): return 'float32'\n"
                "class _U:\n"
                "    class Sequence: pass\n"
                "class keras:\n"
                "    backend = _B\n"
                "    utils = _U\n")
    sys.path.insert(0, d)


def _coo(m):
    m = m.tocoo()
    return np.stack([m.row.astype(np.float64), m.col.astype(np.float64), m.data.astype(np.float64)], axis=1)


def main():
    _install_tf_stub()
    sys.path.insert(0, REF)
    os.chdir(REF)
    from GNN.graph_class import GraphObject
    from GNN.composite_graph_class import CompositeGraphObject

    # ---- MUTAG through the reference loader (patched as described above) ------------------------------------------
    src = open(os.path.join(REF, 'load_MUTAG.py')).read()
    src = src.replace("delimiter=', '", "delimiter=','")
    src = src[:src.index('# HETEROGENEOUS GRAPHS')]
    ns = {}
    exec(compile(src, 'load_MUTAG.py(patched at run time)', 'exec'), ns)
    graphs = ns['graphs']
    assert len(graphs) == 4337

    out = {}
    stats = np.array([[g.nodes.shape[0], g.arcs.shape[0]] for g in graphs])
    out['mutag_stats'] = stats                                  # per-graph (N, E) of all 4337 graphs
    out['mutag_targets'] = np.concatenate([g.targets for g in graphs], axis=0)
    # a checksum of every graph's arcs / nodes so the whole loader is pinned without shipping it twice
    out['mutag_arcs_checksum'] = np.array([float(np.sum(g.arcs * np.arange(1, g.arcs.shape[1] + 1))) for g in graphs])
    out['mutag_nodes_checksum'] = np.array([float(np.sum(g.nodes * np.arange(1, g.nodes.shape[1] + 1)[None, :]
                                                         * np.arange(1, g.nodes.shape[0] + 1)[:, None])) for g in graphs])

    # full arrays for a few single graphs, incl. one with isolated nodes (Q9) and the largest
    iso = [i for i, g in enumerate(graphs) if len(np.unique(g.arcs[:, :2])) != g.nodes.shape[0]]
    picks = [0, 1, 2, int(iso[0]), int(np.argmax(stats[:, 0]))]
    out['single_ids'] = np.array(picks)
    for i in picks:
        for mode in ['sum', 'average', 'normalized']:
            g = graphs[i].copy()
            g.setAggregation(mode)
            p = f'single{i}_{mode}_'
            out[p + 'nodes'], out[p + 'arcs'], out[p + 'targets'] = g.nodes, g.arcs, g.targets
            out[p + 'ArcNode'], out[p + 'Adjacency'], out[p + 'NodeGraph'] = _coo(g.ArcNode), _coo(g.Adjacency), _coo(g.NodeGraph)
            out[p + 'NodeGraph_shape'] = np.array(g.NodeGraph.shape)

    # merged batch: first 32 graphs in file order (BASELINE config C2: N=935, E=1922), all three aggregations, focus g
    for mode in ['sum', 'average', 'normalized']:
        gl = [g.copy() for g in graphs[:32]]
        for g in gl: g.setAggregation(mode)
        m = GraphObject.merge(gl, focus='g', aggregation_mode=mode)
        p = f'merge32_{mode}_'
        out[p + 'nodes'], out[p + 'arcs'], out[p + 'targets'] = m.nodes, m.arcs, m.targets
        out[p + 'set_mask'], out[p + 'output_mask'], out[p + 'sample_weight'] = m.set_mask, m.output_mask, m.sample_weight
        out[p + 'ArcNode'], out[p + 'Adjacency'], out[p + 'NodeGraph'] = _coo(m.ArcNode), _coo(m.Adjacency), _coo(m.NodeGraph)
        out[p + 'NodeGraph_shape'] = np.array(m.NodeGraph.shape)
    merged_stats = []
    for b in range(0, len(graphs), 32):
        m = GraphObject.merge(graphs[b:b + 32], focus='g', aggregation_mode='sum')
        merged_stats.append([m.nodes.shape[0], m.arcs.shape[0]])
    out['merge32_all_stats'] = np.array(merged_stats)           # 136 batches

    # ---- a node-focused toy graph with masks, duplicate arcs and an isolated node -----------------------------------
    rng = np.random.default_rng(7)
    nodes = rng.normal(size=(7, 3))
    arcs = np.array([[0, 1, .5, 1.], [1, 0, .25, 2.], [2, 1, 1., 3.], [3, 1, 2., 4.], [1, 2, 3., 5.], [4, 5, 1., 6.],
                     [5, 4, 2., 7.], [0, 1, .5, 1.], [3, 2, 7., 8.]])       # row 7 duplicates row 0; node 6 isolated
    targets = rng.normal(size=(7, 2))
    set_mask = np.array([1, 1, 1, 0, 1, 1, 1])
    output_mask = np.array([1, 0, 1, 1, 1, 1, 0])
    for mode in ['sum', 'average', 'normalized']:
        g = GraphObject(nodes=nodes, arcs=arcs, targets=targets[:4], focus='n', set_mask=set_mask,
                        output_mask=output_mask, sample_weight=2.5, aggregation_mode=mode)
        p = f'toy_{mode}_'
        out[p + 'arcs_out'] = g.arcs
        out[p + 'ArcNode'], out[p + 'Adjacency'], out[p + 'NodeGraph'] = _coo(g.ArcNode), _coo(g.Adjacency), _coo(g.NodeGraph)
        out[p + 'NodeGraph_shape'] = np.array(g.NodeGraph.shape)
        out[p + 'sample_weight'] = g.sample_weight
    out['toy_nodes'], out['toy_arcs'], out['toy_targets'] = nodes, arcs, targets
    out['toy_set_mask'], out['toy_output_mask'] = set_mask, output_mask

    # ---- composite toy graphs: 3 node types, all four aggregation modes, and a merge --------------------------------
    def composite_toy(seed, n, e):
        r = np.random.default_rng(seed)
        nd = r.normal(size=(n, 4))
        pairs = set()
        while len(pairs) < e:
            a, b = r.integers(0, n, 2)
            if a != b: pairs.add((int(a), int(b)))
        pairs = np.array(sorted(pairs), dtype=float)
        ar = np.concatenate([pairs, r.normal(size=(e, 2))], axis=1)
        types = r.integers(0, 3, n)
        types[:3] = [0, 1, 2]
        tm = np.zeros((n, 3), dtype=bool)
        tm[np.arange(n), types] = True
        tg = r.normal(size=(n, 2))
        return nd, ar, tg, tm

    dim_node_label = (4, 2, 3)
    toys = [composite_toy(11, 9, 20), composite_toy(12, 6, 11)]
    for ti, (nd, ar, tg, tm) in enumerate(toys):
        out[f'ctoy{ti}_nodes'], out[f'ctoy{ti}_arcs'], out[f'ctoy{ti}_targets'], out[f'ctoy{ti}_type_mask'] = nd, ar, tg, tm
    out['ctoy_dim_node_label'] = np.array(dim_node_label)
    for mode in ['sum', 'average', 'normalized', 'composite_average']:
        cgs = [CompositeGraphObject(nodes=nd, arcs=ar, targets=tg, type_mask=tm, dim_node_label=dim_node_label,
                                    focus='n', aggregation_mode=mode) for nd, ar, tg, tm in toys]
        for ti, cg in enumerate(cgs + [CompositeGraphObject.merge(cgs, focus='n', aggregation_mode=mode)]):
            p = f'ctoy{ti}_{mode}_'
            out[p + 'arcs_out'] = cg.arcs
            out[p + 'type_mask_out'] = cg.type_mask
            out[p + 'ArcNode'], out[p + 'Adjacency'] = _coo(cg.ArcNode), _coo(cg.Adjacency)
            for t, ca in enumerate(cg.CompositeAdjacencies):
                out[p + f'CA{t}'] = _coo(ca)

    # ---- transductive re-typing (TransductiveGraphSequencers.py:62-95), seeded numpy RNG -----------------------------------
    from GNN.Sequencers.TransductiveGraphSequencers import TransductiveMultiGraphSequencer
    tn = rng.normal(size=(12, 3))
    ta = np.array([[i, (i + 1) % 12, 1.0] for i in range(12)] + [[i, (i + 5) % 12, 0.5] for i in range(12)])
    tom = np.array([1, 1, 0, 1, 1, 1, 0, 1, 1, 1, 1, 0], dtype=bool)
    tsm = np.array([1, 1, 1, 1, 0, 1, 1, 1, 1, 0, 1, 1], dtype=bool)
    tt = np.eye(2)[rng.integers(0, 2, int(tom.sum()))]
    tg = GraphObject(nodes=tn, arcs=ta, targets=tt, focus='n', set_mask=tsm, output_mask=tom, aggregation_mode='sum')
    out['trans_nodes'], out['trans_arcs'], out['trans_targets'], out['trans_set_mask'], out['trans_output_mask'] = tn, ta, tt, tsm, tom
    for rate in (0.5, 0.3):
        np.random.seed(123)
        cg = TransductiveMultiGraphSequencer.get_transduction(tg, rate, 'n', 'float32')
        p = f'trans_{int(rate * 10)}_'
        out[p + 'nodes'], out[p + 'targets'], out[p + 'type_mask'] = cg.nodes, cg.targets, cg.type_mask
        out[p + 'output_mask'], out[p + 'set_mask'] = cg.output_mask, cg.set_mask

    # ---- files written by the reference's own savers ----------------------------------------------------------------------
    import shutil
    rf = os.path.join(HERE, 'ref_files')
    if os.path.exists(rf): shutil.rmtree(rf)
    os.makedirs(rf)
    graphs[0].save(os.path.join(rf, 'mutag0'))                                   # plain graph: nodes / arcs / targets only
    graphs[1].save_compressed(os.path.join(rf, 'mutag1_compressed'))
    toy = GraphObject(nodes=nodes, arcs=arcs, targets=targets[:4], focus='n', set_mask=set_mask, output_mask=output_mask,
                      sample_weight=2.5, aggregation_mode='average')
    toy.save(os.path.join(rf, 'toy_masks'))                                     # + set_mask / output_mask / sample_weight keys
    toy.savetxt(os.path.join(rf, 'toy_masks_txt'))
    m4 = GraphObject.merge([g.copy() for g in graphs[:4]], focus='g', aggregation_mode='average')
    m4.save(os.path.join(rf, 'merge4'))                                         # + NodeGraph as [value, row, col]
    m4.savetxt(os.path.join(rf, 'merge4_txt'))
    GraphObject.save_dataset(os.path.join(rf, 'dataset_npz'), graphs[2:5])
    GraphObject.save_dataset_txt(os.path.join(rf, 'dataset_txt'), graphs[5:7])
    cg0 = CompositeGraphObject(nodes=toys[0][0], arcs=toys[0][1], targets=toys[0][2], type_mask=toys[0][3],
                               dim_node_label=dim_node_label, focus='n', aggregation_mode='composite_average')
    cg0.save(os.path.join(rf, 'ctoy0'))                                         # + type_mask / dim_node_label keys
    cg0.savetxt(os.path.join(rf, 'ctoy0_txt'))
    out['ref_files_merge4_NodeGraph'] = _coo(m4.NodeGraph)
    out['ref_files_merge4_shape'] = np.array(m4.NodeGraph.shape)
    for name, g in (('mutag0', graphs[0]), ('mutag1', graphs[1]), ('merge4', m4), ('ds2', graphs[2]), ('ds5', graphs[5])):
        out[f'ref_files_{name}_arcs'], out[f'ref_files_{name}_nodes'], out[f'ref_files_{name}_targets'] = g.arcs, g.nodes, g.targets

    path = os.path.join(HERE, 'graph_fixtures.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes;', len(out), 'arrays; isolated-node graphs:', len(iso))


if __name__ == '__main__':
    main()
