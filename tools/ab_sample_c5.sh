#!/bin/bash
# Round 6 (GPU box): the cage samplers per library on C3 (REF + DDGI) and on C5's 3.2 GB of tiles (DDGI), then the HBM counters of the C5 batch.
#   tools/ab_sample_c5.sh <tag>     -> gpurun_out/profiles_out/r06_<tag>_sample_c5.txt
T=${1:-x}
D=$PWD/dynamic-diffuse-global-illumination-minecraft_amd
OUT=$PWD/gpurun_out/sample_c5_$T; mkdir -p $OUT $PWD/gpurun_out/profiles_out
R=$PWD/gpurun_out/profiles_out/r06_${T}_sample_c5.txt
echo "# tools/sample_bench.py per library: 1 440 000 scattered shading points; C3 (REF mode 0, DDGI mode 1), then C5 = 128x64x128 probes, DDGI mode: 3.2 GB of tiles" > $R
for rep in 1 2; do for lib in $(cd $D; ls libddgi_probe*.so | grep -v prof); do
  echo "== $lib, C3 (rep $rep)" >> $R; DDGI_LIB=$D/$lib python tools/sample_bench.py 2>/dev/null | grep "^mode" >> $R
  echo "== $lib, C5 DDGI (rep $rep)" >> $R; DDGI_LIB=$D/$lib SAMPLE_WORKLOAD=c5 SAMPLE_MODES=1 python tools/sample_bench.py 2>/dev/null | grep "^mode" >> $R
done; done
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/tools/sample_bench.py"
export SAMPLE_WORKLOAD=c5 SAMPLE_MODES=1
rocprofv3 --kernel-trace --stats -d $OUT -o kt --output-format csv -- $CMD > $OUT/kt.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT -o fetch --output-format csv -- $CMD > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT -o write --output-format csv -- $CMD > $OUT/write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $OUT -o tcc --output-format csv -- $CMD > $OUT/tcc.log 2>&1
python3 - <<PY >> $R
import csv, collections
print("# C5 DDGI batch under rocprofv3 (default library): kernel times, then counters per launch (FETCH_SIZE / WRITE_SIZE in the counters' own units — see MI355X_MICROARCH.md for the gfx950 correction)")
rows = list(csv.DictReader(open("$OUT/kt_kernel_trace.csv")))
agg = collections.OrderedDict()
for r in rows:
    if "sample" in r["Kernel_Name"]:
        agg.setdefault(r["Kernel_Name"].split("(")[0], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("%-48s %6s %10s %10s" % ("kernel", "calls", "avg_us", "min_us"))
for k, v in agg.items(): print("%-48s %6d %10.2f %10.2f" % (k[:48], len(v), sum(v) / len(v), min(v)))
for f in ("fetch", "write", "tcc"):
    a = collections.defaultdict(list)
    try:
        for r in csv.DictReader(open("$OUT/" + f + "_counter_collection.csv")):
            if "sample" in r["Kernel_Name"]: a[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    except Exception as e:
        print("#", f, "no data:", e); continue
    for (k, c), v in sorted(a.items()): print("%-40s %-26s %14.6g  (%d launches)" % (k[:40], c, sum(v) / len(v), len(v)))
PY
cat $R
