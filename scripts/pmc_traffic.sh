#!/bin/bash
# usage: pmc_traffic.sh TAG [bench args]: FETCH_SIZE and WRITE_SIZE passes (separate runs, as the guide prescribes)
export TMPDIR=/tmp; R=$PWD; tag=$1; shift
BARGS="$*"
mkdir -p gpurun_out/prof_$tag
run() { name=$1; shift; timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/prof_$tag/$name -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-elems 0 $BARGS > gpurun_out/prof_$tag/$name.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/prof_$tag/$name.log; return 1; }; }
run fetch FETCH_SIZE && run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python3 - <<PY
import csv, glob, collections
for p in ("fetch", "write"):
    acc = collections.defaultdict(list)
    for fn in glob.glob("gpurun_out/prof_$tag/%s/*/*counter_collection.csv" % p):
        for r in csv.DictReader(open(fn)):
            if "assemble" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(p, k, "n=%d mean=%.6g" % (len(v), sum(v) / len(v)))
PY
