from .gcn_nssc import GCNInfer, GCNSampling
from .graphsage_nssc import GraphSageSampling
