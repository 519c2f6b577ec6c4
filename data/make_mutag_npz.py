#!/usr/bin/env python3
"""Pack the raw TU-Mutagenicity text files shipped with the reference (`/root/reference/MUTAG_raw/*.txt`, the *input
data* of BASELINE configs C1/C2) into one compact integer archive, `data/mutagenicity.npz`, so the dataset can travel
to the GPU box (where /root/reference does not exist). Data only: arrays are stored exactly as they appear in the files
(1-based node ids, file order); all parsing logic of the reference loader is re-implemented in `gnnkeras_amd/load_MUTAG.py`.

    python data/make_mutag_npz.py
"""
import os
import numpy as np

RAW = '/root/reference/MUTAG_raw/'
HERE = os.path.dirname(os.path.abspath(__file__))

edges = np.loadtxt(RAW + 'Mutagenicity_edges.txt', dtype=np.int32, delimiter=',')
np.savez_compressed(os.path.join(HERE, 'mutagenicity.npz'),
                    edges=edges,
                    edge_labels=np.loadtxt(RAW + 'Mutagenicity_edge_labels.txt', dtype=np.int8),
                    node_labels=np.loadtxt(RAW + 'Mutagenicity_node_labels.txt', dtype=np.int8),
                    graph_indicator=np.loadtxt(RAW + 'Mutagenicity_graph_indicator.txt', dtype=np.int32),
                    graph_labels=np.loadtxt(RAW + 'Mutagenicity_graph_labels.txt', dtype=np.int8))
print(os.path.getsize(os.path.join(HERE, 'mutagenicity.npz')), 'bytes')
