#!/bin/bash
O=gpurun_out/sw; mkdir -p $O
T=tests/test_gpu_parity.py
for i in 1 2; do timeout 600 python -m pytest $T -x -q -k test_bench_short_window > $O/alone$i.log 2>&1; tail -1 $O/alone$i.log; done
timeout 1500 python -m pytest tests -m gpu -x -q > $O/suite_new.log 2>&1; tail -2 $O/suite_new.log; grep -n "AssertionError" $O/suite_new.log | head -3
PG_MISSQ_NO_DIRECT=2 timeout 1500 python -m pytest tests -m gpu -x -q > $O/suite_old_scatter.log 2>&1; tail -2 $O/suite_old_scatter.log; grep -n "AssertionError" $O/suite_old_scatter.log | head -3
