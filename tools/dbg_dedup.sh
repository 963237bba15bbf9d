#!/bin/bash
O=gpurun_out/fallbacks; mkdir -p $O
K="fetch or dedup or trainer or virtual or hardware_queue or reddit_width or config3"
PG_MISSQ_HSA_COPY=0 timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "$K" > $O/nohsa.log 2>&1; echo "PG_MISSQ_HSA_COPY=0: $(tail -1 $O/nohsa.log)"
PG_MISSQ_NO_DIRECT=1 timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "$K" > $O/nodirect.log 2>&1; echo "PG_MISSQ_NO_DIRECT=1: $(tail -1 $O/nodirect.log)"
PG_MISSQ_HOST_WAIT=1 timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "$K" > $O/hostwait.log 2>&1; echo "PG_MISSQ_HOST_WAIT=1: $(tail -1 $O/hostwait.log)"
PG_DEDUP_MISSES=0 timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "fetch or trainer or virtual" > $O/nodedup.log 2>&1; echo "PG_DEDUP_MISSES=0: $(tail -1 $O/nodedup.log)"
