"""pagraph_amd — MI355X-native PaGraph minibatch hot path (feature cache + gather,
neighbour sampling, block aggregation, dg partition) behind the reference's trainer API.
Everything compute goes through libpagraph_hip.so (see include/pagraph_hip.h); there
is no CPU fallback."""
__version__ = "0.1.0"
