from .nodeflow import DeviceGraph, NodeFlow
from .sampler import NeighborSampler
