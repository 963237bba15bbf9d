mkdir -p gpurun_out/r05g; O=$PWD/gpurun_out/r05g; R=$PWD
export TMPDIR=/tmp
S="--skip-microbench --skip-cpu-baseline --skip-opt-hit --skip-reference-equivalent --no-configs"
prof() {  # name, env..., -- args
  name=$1; shift
  ( cd /tmp && env "$@" timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o b -- python $R/bench.py $S $ARGS > $O/bench_$name.json 2> /tmp/prof_$name.log )
  cp /tmp/prof_$name/*kernel_stats.csv $O/kernel_stats_$name.csv
  python tools/trace_seq.py /tmp/prof_$name/b_kernel_trace.csv > $O/step_sequence_$name.txt 2>&1
  echo "== $name"; cat $O/step_sequence_$name.txt; head -14 $O/kernel_stats_$name.csv | cut -c1-150
}
ARGS="--cache-ratio 1.0"
prof fc_nat1_gate1 PG_NATIVE_PREPARE=1 PG_PHASE_GATE=1
prof fc_nat0_gate0 PG_NATIVE_PREPARE=0 PG_PHASE_GATE=0
ARGS="--vertices 100000000 --edges 1000000000 --steps 400"
prof scale_flags PG_X=1
prof scale_noflags PG_SAMPLER_NO_UNIT_FLAGS=1
