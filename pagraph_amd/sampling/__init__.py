from .nodeflow import DeviceGraph, NodeFlow
from .sampler import NeighborSampler
from .full import full_neighbor_nodeflow
