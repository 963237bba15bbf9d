export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/p3
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcdg_$c -o p -- python "$R/tools/exp_dg_gpu.py" 10M 4 2 > /tmp/pmcdg_$c.log 2>&1 )
done
python - <<'EOF'
import csv, collections
def tot(path, name):
    s = 0.0; n = 0; dur = 0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name and "k_dg_expand" in r["Kernel_Name"]:
            s += float(r["Counter_Value"]); n += 1; dur += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return s, n, dur
f, n, d = tot("/tmp/pmcdg_FETCH_SIZE/p_counter_collection.csv", "FETCH_SIZE")
w, n2, d2 = tot("/tmp/pmcdg_WRITE_SIZE/p_counter_collection.csv", "WRITE_SIZE")
print("k_dg_expand launches", n, "fetch GB (x2 corrected)", 2 * f * 1024 / 1e9, "write GB", w * 1024 / 1e9, "kernel seconds", d / 1e9, d2 / 1e9)
# entries of the TRAIN vertices' two-hop multisets on the 10M / 100M graph: 3.085e10 (tools/exp_dg_multiset_shape.py); one walk
# each since the generation-tagged bitmaps (the few multisets whose put-aside list overflows are walked twice)
print("per multiset entry (3.085e10, one walk): fetch B", 2 * f * 1024 / 3.085e10, "write B", w * 1024 / 3.085e10)
EOF
