#!/usr/bin/env python3
"""BASELINE config C1 on the device: the reference's `starter.py` experiment (MUTAG / Mutagenicity graph-focused binary
classification, batch 32) with the reference's hyper-parameters, run through this package's API.

    python examples/mutag_starter.py [--epochs 10] [--lgnn]

Importing this module (instead of running it) builds the same module-level objects a user of the reference's starter
gets: `gnn`, `lgnn`, `gTr_Sequencer`, `gVa_Sequencer`, `gTe_Sequencer` (reference README.md:64).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

from gnnkeras_amd.load_MUTAG import load_graphs
from gnnkeras_amd.Models.GNN import GNNgraphBased
from gnnkeras_amd.Models.LGNN import LGNN
from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
from gnnkeras_amd.Models.training import Adam
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer

# options of the reference's starter.py:16-47
aggregation_mode, focus = 'average', 'g'
dim_state, max_iter, state_threshold = 0, 5, 0.01
layers, get_state, get_output, training_mode = 3, True, True, 'serial'
batch_size, learning_rate, seed = 32, 0.01, 0

graphs = load_graphs()
for g in graphs: g.setAggregation(aggregation_mode)
np.random.default_rng(seed).shuffle(graphs)
gTr, gTe, gVa = graphs[:-1500], graphs[-1500:-750], graphs[-750:]                 # starter.py:63-66
L, A, T = int(gTr[0].DIM_NODE_LABEL[0]), gTr[0].DIM_ARC_LABEL, gTr[0].DIM_TARGET


def nets(layer):
    inp, lay = get_inout_dims('state', L, A, T, focus, dim_state, layer=layer, get_state=get_state, get_output=get_output)
    ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=seed + 10 * layer, name=f'State_{layer}')
    inp, lay = get_inout_dims('output', L, A, T, focus, dim_state, layer=layer, get_state=get_state, get_output=get_output)
    no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=seed + 10 * layer + 1, name=f'Out_{layer}')
    return ns, no


gnn = GNNgraphBased(*nets(0), dim_state, max_iter, state_threshold)
gnn.compile(optimizer=Adam(learning_rate), loss='categorical_crossentropy', average_st_grads=False, metrics=['accuracy'],
            run_eagerly=True)
lgnn = LGNN([GNNgraphBased(*nets(i), dim_state, max_iter, state_threshold) for i in range(layers)], get_state, get_output)
lgnn.compile(optimizer=Adam(learning_rate), loss='categorical_crossentropy', average_st_grads=True, metrics=['accuracy'],
             run_eagerly=True, training_mode=training_mode)

gTr_Sequencer = MultiGraphSequencer(gTr, focus, aggregation_mode, batch_size)
gVa_Sequencer = MultiGraphSequencer(gVa, focus, aggregation_mode, batch_size)
gTe_Sequencer = MultiGraphSequencer(gTe, focus, aggregation_mode, batch_size)

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--epochs', type=int, default=10)
    ap.add_argument('--lgnn', action='store_true')
    args = ap.parse_args()
    np.random.seed(seed)
    model = lgnn if args.lgnn else gnn
    model.fit(gTr_Sequencer, epochs=args.epochs, validation_data=gVa_Sequencer)
    print('test :', model.evaluate(gTe_Sequencer, return_dict=True))
