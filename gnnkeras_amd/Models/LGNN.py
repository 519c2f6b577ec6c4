"""Layered GNN (LGNN): a stack of GNNs, layer i+1 sees the original labels extended with the state and/or output of
layer i. Host-side orchestration over the device loop — mirror of the reference's `GNN/Models/LGNN.py`
(constructor, `compile(training_mode=...)`, `call`, `Loop`, `update_graph`, `train_step`, serial `fit`, save / load).

Every layer's message-passing loop is the native one (`gnn_loop_forward`, or the training tape of
`Models/training.py`); what this file adds is the label plumbing between layers and, for the joint training modes,
the chaining of gradients through `update_graph`:

    nodes_{i+1} = [ state_i (get_state) | out_i scattered on the mask (get_output, node / graph focus) | nodes_0 ]

so d loss / d nodes_{i+1} splits into an extra gradient on layer i's final state and on its per-node outputs
(reference: one eager GradientTape over all layers, `LGNN.py:252-287`).

Reference quirk kept on purpose (it defines the arithmetic): for arc-focused stacks with `get_output`, `update_graph`
concatenates the outputs *in front of the id columns* of `arcs` (`LGNN.py:210`), so the next layer's arc "labels"
`arcs[:, 2:]` are `[out[:, 2:] | src id | dst id | labels]`.
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch

from .GNN import GNNnodeBased, GNNarcBased, GNNgraphBased, _LoopModel, _metric_fn, _squeeze_last
from .training import LoopTrainer


class LGNN(_LoopModel):
    """Layered GNN for node-, arc- or graph-focused problems (reference LGNN.py:11-362)."""
    _gnn_classes = {"node": GNNnodeBased, "arc": GNNarcBased, "graph": GNNgraphBased}

    def __init__(self, gnns: list, get_state: bool, get_output: bool) -> None:
        assert get_state or get_output
        assert len(set([type(i) for i in gnns])) == 1
        self.GNN_CLASS = type(gnns[0])
        self.gnns = gnns
        self.LAYERS = len(gnns)
        self.get_state = bool(get_state)
        self.get_output = bool(get_output)
        self.training_mode = None
        self._engine_init()

    # the class <-> name maps of the reference (`__gnnClass__`, `__gnnClassLoader__`)
    @classmethod
    def _class_name(cls, klass):
        return {v: k for k, v in cls._gnn_classes.items()}[klass]

    @property
    def _focus(self):
        return self.gnns[0]._focus

    def copy(self, copy_weights: bool = True):
        config = self.get_config()
        config["gnns"] = [i.copy(copy_weights=copy_weights) for i in config["gnns"]]
        return self.from_config(config)

    def get_config(self):
        return {"gnns": self.gnns, "get_state": self.get_state, "get_output": self.get_output}

    @classmethod
    def from_config(cls, config, **kwargs):
        return cls(**config)

    def __repr__(self):
        return f"LGNN(type={self._class_name(self.GNN_CLASS)}, layers={self.LAYERS}, " \
               f"get_state={self.get_state}, get_output={self.get_output}, " \
               f"mode={self.training_mode}, avg={self.average_st_grads})"

    __str__ = __repr__

    def save(self, path: str, *args, **kwargs):
        """`<path>/GNN{i}/` per layer + `<path>/config.json` (reference LGNN.py:83-101)."""
        if path[-1] != '/': path += '/'
        config = self.get_config()
        config["gnn_class"] = self._class_name(self.GNN_CLASS)
        for i, gnn in enumerate(config.pop("gnns")): gnn.save(f'{path}GNN{i}/', **kwargs)
        with open(f'{path}config.json', 'w') as json_file: json.dump(config, json_file)

    @classmethod
    def load(cls, path: str):
        if path[-1] != '/': path += '/'
        with open(f'{path}config.json', 'r') as read_file: config = json.loads(read_file.read())
        gnn_class = cls._gnn_classes[config.pop('gnn_class')]
        dirs = sorted((d for d in os.listdir(path) if os.path.isdir(f'{path}{d}')), key=lambda d: int(d[3:]))
        return cls(gnns=[gnn_class.load(f'{path}{d}') for d in dirs], **config)

    def compile(self, *args, training_mode: str = 'parallel', average_st_grads: bool = False, **kwargs):
        """`training_mode` in 'serial' (layers trained one after another), 'parallel' (loss = mean of the layers' losses),
        'residual' (loss of the mean of the layers' outputs) — reference LGNN.py:133-152."""
        if training_mode not in ('serial', 'parallel', 'residual'): raise ValueError('unknown training_mode')
        super().compile(*args, average_st_grads=average_st_grads, **kwargs)
        for gnn in self.gnns: gnn.compile(*args, average_st_grads=average_st_grads, **kwargs)
        self.training_mode = training_mode

    # ---- forward -----------------------------------------------------------------------------------------------------
    process_inputs = staticmethod(GNNnodeBased.process_inputs)

    def call(self, inputs, training: bool = False, mask=None):
        inputs = self.process_inputs(inputs)
        k, state, out = self.Loop(*inputs, training=training)
        if training: return k, state, out
        return out[-1]

    def update_graph(self, nodes, arcs, dim_node_label, set_mask, output_mask, state, output):
        """New (nodes, arcs, dim_node_label) with the state / output of a layer merged into the ORIGINAL labels
        (reference LGNN.py:175-214). Works on torch tensors (device) or numpy arrays (serial fit updates GraphObjects)."""
        as_np = not isinstance(nodes, torch.Tensor)
        t = (lambda x: torch.as_tensor(np.asarray(x))) if as_np else (lambda x: x)
        nodes, arcs, state, output = t(nodes).float(), t(arcs).float(), t(state).float(), t(output).float()
        set_mask, output_mask = _squeeze_last(t(set_mask)).bool(), _squeeze_last(t(output_mask)).bool()
        nodeplus, arcplus = [], []
        if self.get_state: nodeplus.append(state.to(nodes.device))
        if self.get_output:
            mask = torch.logical_and(set_mask, output_mask).to(nodes.device)
            out = torch.zeros((len(mask), output.shape[1]), dtype=torch.float32, device=nodes.device)
            out[mask] = output.to(nodes.device)
            (arcplus if self.GNN_CLASS is self._gnn_classes['arc'] else nodeplus).append(out)
        plus = sum(x.shape[1] for x in nodeplus)
        nodes = torch.cat(nodeplus + [nodes], dim=1)
        arcs = torch.cat(arcplus + [arcs], dim=1)
        dim_node_label = (np.asarray(dim_node_label) if as_np else dim_node_label) + plus
        if as_np: return nodes.numpy(), arcs.numpy(), dim_node_label
        return nodes, arcs, dim_node_label

    def _layer_inputs(self, nodes, arcs, dim_node_label, constant_inputs):
        return [nodes, arcs, dim_node_label] + list(constant_inputs)

    def Loop(self, nodes, arcs, dim_node_label, set_mask, output_mask, adjacency, arcnode, nodegraph,
             training: bool = False, *, state0=None, seed=None):
        """Lists (K, states, outs), one entry per layer (reference LGNN.py:217-249). `state0`: optional list of
        per-layer initial states."""
        constant_inputs = [set_mask, output_mask, adjacency, arcnode, nodegraph]
        nodes_0, arcs_0 = nodes, arcs
        s0 = state0 if state0 is not None else [None] * self.LAYERS
        K, states, outs = [], [], []
        graph_based = self.GNN_CLASS is self._gnn_classes['graph']
        for idx, gnn in enumerate(self.gnns[:-1]):
            k, state, out = gnn.Loop(nodes, arcs, dim_node_label, *constant_inputs, training=training, state0=s0[idx],
                                     seed=seed, node_level=True)
            K.append(k); states.append(state)
            outs.append(self._pool(nodegraph, out) if graph_based else out)
            nodes, arcs, dim_node_label = self.update_graph(nodes_0, arcs_0, dim_node_label, set_mask, output_mask, state, out)
        k, state, out = self.gnns[-1].Loop(nodes, arcs, dim_node_label, *constant_inputs, training=training,
                                           state0=s0[-1], seed=seed)
        return K + [k], states + [state], outs + [out]

    @staticmethod
    def _pool(nodegraph, out_nodes):
        """NodeGraph^T . out (per-graph mean of node outputs) on the device: `torch.ops.gnnkeras.pool`."""
        from .. import ops
        from ..sparse import SparseMatrix
        return ops.pool(SparseMatrix.from_triple(nodegraph).device_csr(out_nodes.device), out_nodes.to(torch.float32).contiguous())

    # ---- joint training (parallel / residual) ----------------------------------------------------------------------------
    def _layer_x(self, x, nodes, arcs, dim_node_label):
        return [nodes, arcs, dim_node_label] + list(x[3:])

    def train_step(self, data, *, state0=None, seed=None, apply=True):
        """One joint optimisation step of all layers (reference LGNN.py:252-287): 'parallel' = mean of the per-layer
        losses, 'residual' = loss of the mean output. Gradients flow from layer i+1 into layer i through the labels
        built by `update_graph`."""
        if self.training_mode == 'serial':
            raise RuntimeError("training_mode 'serial' trains layer by layer: use fit()")
        if self.loss is None: raise RuntimeError('compile() the model with a loss before fit() / train_step()')
        x, y, sample_weight = data
        if y is None: raise TypeError('Target data is missing. Your model was compiled with `loss` '
                                      'argument and so expects targets to be passed in `fit()`.')
        x = list(x)
        i_set = 4 if len(x) == 10 else 3                           # composite lists carry type_mask at [3]
        nodes_0, arcs_0, dim0, set_mask, output_mask = x[0], x[1], x[2], x[i_set], x[i_set + 1]
        s0 = state0 if state0 is not None else [None] * self.LAYERS
        trainers = [LoopTrainer(g) for g in self.gnns]
        for g in self.gnns: g.loss = self.loss
        graph_based = self.GNN_CLASS is self._gnn_classes['graph']
        arc_based = self.GNN_CLASS is self._gnn_classes['arc']
        tapes, outs = [], []
        nodes, arcs, dnl = nodes_0, arcs_0, dim0
        for i, tr in enumerate(trainers):
            last = i == self.LAYERS - 1
            tp = tr.forward(self._layer_x(x, nodes, arcs, dnl), state0=s0[i], seed=seed, node_level=not last)
            tapes.append(tp)
            if last: outs.append(tp.y_pred)
            else:
                outs.append(self._pool(x[-1], tp.out_nodes) if graph_based else tp.out_nodes)
                nodes, arcs, dnl = self.update_graph(nodes_0, arcs_0, dnl, set_mask, output_mask, tp.state, tp.out_nodes)
        # loss and its gradient w.r.t. every layer's task-level output
        Lyr = self.LAYERS
        if self.training_mode == 'parallel':
            parts = [trainers[i].loss_and_grad(tapes[i], outs[i], y, sample_weight) for i in range(Lyr)]
            loss = sum(pl[0] for pl in parts) / Lyr
            dpreds = [pl[1] / Lyr for pl in parts]
        else:
            mean_out = sum(outs) / Lyr
            loss, dmean = trainers[-1].loss_and_grad(tapes[-1], mean_out, y, sample_weight)
            dpreds = [dmean / Lyr for _ in range(Lyr)]
        # backward, last layer first; the label gradient of layer i+1 feeds layer i
        d_state_extra, d_out_extra = None, None
        for i in range(Lyr - 1, -1, -1):
            tp, tr = tapes[i], trainers[i]
            pooled_here = graph_based                              # every layer's task output is pooled for graph focus
            G = tr.pool_backward(tp, dpreds[i]) if pooled_here else dpreds[i].clone()
            if d_out_extra is not None: G = G + d_out_extra
            arc_out = arc_based and self.get_output and i > 0        # the layer below wrote its output into the ARC labels
            d_nodes = tr.backward(tp, G.contiguous(), d_state_extra=d_state_extra, want_label_grads=i > 0,
                                  want_arc_label_grads=arc_out)
            d_state_extra = d_out_extra = None
            if i > 0:
                prev = tapes[i - 1]
                col = 0
                if self.get_state:
                    d_state_extra = d_nodes[:, :prev.S].contiguous(); col = prev.S
                if self.get_output and not arc_based:
                    d_out_extra = d_nodes[:, col:col + prev.T].index_select(0, prev.out_index.long()).contiguous()
                elif arc_out:
                    # update_graph PREPENDS the T output columns to the arcs matrix (reference LGNN.py:209: `concat([arcplus,
                    # arcs])`), so the model's label view arcs[:, 2:] starts at output column 2: columns 0 and 1 sit where the
                    # arc ids used to be and are never read; output column 2 + j is arc-label column j
                    d_out_extra = torch.zeros((prev.M, prev.T), dtype=torch.float32, device=tp.dev)
                    if prev.T > 2:
                        d_out_extra[:, 2:] = tp.d_arc_labels[:, :prev.T - 2].index_select(0, prev.out_index.long())
        for tp, tr in zip(tapes, trainers): tr.finish(tp, apply=False)
        if apply:
            gv = [pair for tp in tapes for pair in LoopTrainer.grads_and_vars(tp)]
            self._optimizer_obj().apply_gradients(gv)
        self._last_tapes = tapes
        out = {'loss': loss, 'k': [tp.k for tp in tapes]}
        yd = y.to(outs[-1].device)
        sw = torch.ones(yd.shape[0], device=yd.device) if sample_weight is None else sample_weight.to(yd.device)
        for mtr in self.metrics_spec:
            n, f = _metric_fn(mtr, yd.shape[-1])
            out[n] = (f(yd, outs[-1]) * sw).sum() / sw.sum()
        return out

    # ---- fit: serial mode trains the layers one after another (reference LGNN.py:290-362) -----------------------------------
    def fit(self, sequencer, epochs: int = 1, validation_data=None, verbose: int = 1, **kwargs):
        """'parallel' / 'residual': the usual loop.  'serial' (reference LGNN.py:290-362): the layers are trained one after
        another, each on the graphs relabelled with its predecessor's states / outputs; `callbacks`, when given, is a list of
        LAYERS callback lists - entry i goes to layer i's fit (reference :299-303)."""
        if self.training_mode != 'serial':
            return super().fit(sequencer, epochs=epochs, validation_data=validation_data, verbose=verbose, **kwargs)
        callbacks = kwargs.pop('callbacks', None)
        if callbacks is None: callbacks = [[] for _ in range(self.LAYERS)]
        assert len(callbacks) == self.LAYERS
        histories = []
        # (the reference copies a sequencer wherever it hands one on, LGNN.py:303-313; a layer's fit() and the propagation below re-batch and
        # shuffle but never edit a graph, so a second sequencer over the same graph objects does - only the relabelled graphs are copies)
        view = lambda s: s._view() if hasattr(s, '_view') else s.copy()
        train_t0, valid_t0 = sequencer, validation_data
        training_sequence = view(train_t0)
        valid_sequence = view(valid_t0) if valid_t0 is not None else None

        def propagate(gnn, seq_now, seq_t0):
            """states / outputs of every single graph (batch size 1, training-mode forward as in the reference), merged
            into the t0 graphs' labels."""
            seq_now.shuffle = False
            seq_now.set_batch_size(1)
            results = [gnn.Loop(*gnn.process_inputs(seq_now[i][0]), training=True, node_level=True) for i in range(len(seq_now))]
            new_seq = seq_t0.copy()
            for g, (_, s, o) in zip(new_seq.data, results):
                n, a, l = self.update_graph(g.nodes, g.arcs, g.DIM_NODE_LABEL, g.set_mask, g.output_mask,
                                            s.cpu().numpy(), o.cpu().numpy())
                g.nodes, g.arcs, g.DIM_NODE_LABEL = n, a, l
            return new_seq

        for idx, gnn in enumerate(self.gnns[:-1]):
            if verbose: print(f'\\n\\n --- GNN {idx + 1}/{self.LAYERS} ---')
            histories.append(gnn.fit(view(training_sequence), epochs=epochs, verbose=verbose, callbacks=callbacks[idx],
                                     validation_data=view(valid_sequence) if valid_sequence is not None else None, **kwargs))
            training_sequence = propagate(gnn, training_sequence, train_t0)
            if valid_sequence is not None: valid_sequence = propagate(gnn, valid_sequence, valid_t0)
        if verbose: print(f'\\n\\n --- GNN {self.LAYERS}/{self.LAYERS} ---')
        histories.append(self.gnns[-1].fit(view(training_sequence), epochs=epochs, verbose=verbose, callbacks=callbacks[-1],
                                           validation_data=view(valid_sequence) if valid_sequence is not None else None, **kwargs))
        self.history = histories
        return histories
