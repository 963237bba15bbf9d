"""pagraph_amd — MI355X-native PaGraph minibatch hot path (feature cache + gather,
neighbour sampling, block aggregation, dg partition) behind the reference's trainer API.
Everything compute goes through libpagraph_hip.so (see include/pagraph_hip.h); there
is no CPU fallback."""
__version__ = "0.1.0"
import os as _os

# The HIP runtime multiplexes the streams of one priority class onto at most GPU_MAX_HW_QUEUES hardware queues (default 4 per
# class). Measured on MI355X / ROCm 7.0 (round 6, profiles/r06/hw_queues*.txt): once the process owns more than about six of
# them — the pipeline's compute / sampler / load / copy streams plus torch's default stream are five; torch.distributed's NCCL
# stream and a communication stream make seven — EVERY small kernel of the side streams takes ~45 us longer and the training
# step doubles (0.10 -> 0.20-0.32 ms). Two queues per class (the pipeline never has more than two busy streams in one class)
# keep the step where it was, with or without the extra streams. Read by the runtime when it initialises, i.e. at the
# process's first HIP call: import pagraph_amd (or set the variable) before touching torch.cuda. An explicit setting wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
