#!/bin/bash
# usage: tools/bench_ms.sh [bench args...] -> "ms_per_step epoch_s"
python bench.py --skip-cpu-baseline --skip-opt-hit --skip-microbench "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.4f  epoch %.4f s  host-issue %.4f ms/step' % (d['ms_per_step'], d['value'], d.get('host_issue_ms_per_step', -1)))"
