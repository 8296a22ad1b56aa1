#!/bin/bash
# usage: r03_profiles.sh TAG KEY [bench args]   -> gpurun_out/r03_prof/TAG/{bench.json,stats,timed_kernel_stats.csv,fetch,write}
# One bench configuration's evidence: the JSON line, rocprofv3 kernel-trace statistics (raw, and over the timed launches
# only), and the two HBM traffic counter passes (FETCH_SIZE, WRITE_SIZE), every profiler pass in a run of its own.
# KEY: the traffic.json key bench.py builds (e.g. cfg3:16000000:overwrite:full:K)
export TMPDIR=/tmp; R=$PWD; tag=$1; key=$2; shift; shift; O=$R/gpurun_out/r03_prof/$tag; mkdir -p $O
B="python3 $R/bench.py --cpu-elems 0 --no-f-int-only $*"
W=3
(cd /tmp && timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --steps 8 --warmup $W > $O/stats.log 2>&1) || { echo "$tag: stats pass failed"; tail -3 $O/stats.log; exit 1; }
python3 scripts/rocprof_timed_stats.py $O/stats $W > $O/timed_kernel_stats.csv
(cd /tmp && timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B --steps 3 --warmup 1 > $O/fetch.log 2>&1) || { echo "$tag: fetch pass failed"; tail -3 $O/fetch.log; exit 1; }
(cd /tmp && timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $B --steps 3 --warmup 1 > $O/write.log 2>&1) || { echo "$tag: write pass failed"; tail -3 $O/write.log; exit 1; }
python3 scripts/make_traffic.py $O "$key" > $O/traffic.txt || { echo "$tag: traffic failed"; exit 1; }
cat $O/traffic.txt
# the bench line last: it picks up the traffic just measured with this very library
timeout -k 10 400 $B --steps 10 --warmup 3 2> $O/bench.err | tail -1 > $O/bench.json || { echo "$tag: bench failed"; tail -3 $O/bench.err; exit 1; }
python3 -c "import json; d=json.load(open('$O/bench.json')); r=d['roofline']; print('$tag: kernel_ms %.3f frac %.4f value %.4g traffic %s (%s)' % (r['kernel_ms'], r['frac'], d['value'], r['traffic'], r.get('traffic_source')))"
head -3 $O/timed_kernel_stats.csv | cut -c1-200
