#!/bin/bash
O=gpurun_out/sage; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "deferred or graphed_trainer or virtual_layer0 or models_vs_reference or reddit_width or two_rank" > $O/tests.log 2>&1; tail -4 $O/tests.log
pick='import json,sys,statistics as st; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=d["ms_per_step_windows"]; q=d["miss_queue"] or {}; print(round(d["ms_per_step"],4), "median window", round(st.median(w),4), "gather us", round(q.get("us_cpu_gather",0)))'
B="python bench.py --steps 1084 --warmup 20 --skip-cpu-baseline --skip-microbench --skip-opt-hit --skip-reference-equivalent"
echo "== gcn"; timeout 300 $B 2> $O/last.err | python -c "$pick" || tail -5 $O/last.err
echo "== graphsage 30%"; timeout 300 $B --model graphsage 2> $O/last.err | python -c "$pick" || tail -5 $O/last.err
echo "== graphsage full cache"; timeout 300 $B --model graphsage --cache-ratio 1.0 2> $O/last.err | python -c "$pick" || tail -5 $O/last.err
echo "== graphsage 30% unfused partials"; PG_NO_DEFER_SAGE=1 timeout 300 $B --model graphsage 2> $O/last.err | python -c "$pick" || tail -5 $O/last.err
cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_gs -o b -- python $GRAFT_REPO_ROOT/bench.py --model graphsage --skip-cpu-baseline --skip-opt-hit --skip-reference-equivalent --skip-microbench > /tmp/prof_gs.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/trace_seq.py /tmp/prof_gs/b_kernel_trace.csv > $O/graphsage_step_sequence.txt 2>&1; tail -3 $O/graphsage_step_sequence.txt
