"""gnnkeras_amd — MI355X-native convergent message-passing loop behind GNNkeras' GraphObject / Sequencer / GNN API."""
from .sparse import SparseMatrix, SparseTriple, default_device
from .graph_class import GraphObject, GraphTensor
from .composite_graph_class import CompositeGraphObject, CompositeGraphTensor

__all__ = ['SparseMatrix', 'SparseTriple', 'default_device', 'GraphObject', 'GraphTensor',
           'CompositeGraphObject', 'CompositeGraphTensor']
