#!/bin/bash
# A second GraphedTrainer in one process ran into 3 s miss-queue time-outs (bench's reference-equivalent leg after the
# oracle-hit leg; the last test of a long pytest session): the copy stream shared a hardware queue with the sampler / load
# stream of its own pipeline (see pg_missq_create). Re-check: bench with every leg, GCN and GraphSAGE, twice.
O=gpurun_out/dbg2
mkdir -p $O
export PG_MISSQ_DEBUG=1
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],4), round(d["reference_equivalent"]["ms_per_step"],4), d["ms_per_step_windows"][:8], d["miss_queue"]["timed_region"])'
B0="python bench.py --steps 200 --warmup 20 --skip-cpu-baseline --skip-microbench"
run() { echo "== $1"; shift; env "$@" timeout 300 $B0 $EXTRA 2> $O/last.err | python -c "$pick" || tail -5 $O/last.err; }
run "gcn" X=1
run "gcn again" X=1
run "gcn ring 4 (old depth)" X=1 
EXTRA="--ring 4"; run "gcn ring 4" X=1
EXTRA="--ring 6"; run "gcn ring 6" X=1
EXTRA="--model graphsage"; run "graphsage" X=1
run "graphsage again" X=1
EXTRA=""; echo "== driver invocation"; python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python -c "$pick"
python bench.py 2>/dev/null | python -c "$pick"
