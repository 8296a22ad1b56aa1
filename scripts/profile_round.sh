#!/bin/bash
# usage: scripts/profile_round.sh TAG   -> gpurun_out/prof_TAG/{stats,fetch,write,calib_fetch,calib_write}
# kernel-trace statistics and HBM traffic counters of the default bench command, each in a run of its own.
export TMPDIR=/tmp; R=$PWD; tag=$1; O=$R/gpurun_out/prof_$tag; mkdir -p $O
B="python3 $R/bench.py --steps 10 --warmup 2 --cpu-elems 0"
timeout -s KILL 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/stats.log 2>&1 && \
timeout -s KILL 280 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > $O/fetch.log 2>&1 && \
timeout -s KILL 280 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > $O/write.log 2>&1 && \
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/calib_fetch -- $R/scripts/calib/fetch_calib > $O/calib_fetch.log 2>&1 && \
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/calib_write -- $R/scripts/calib/fetch_calib > $O/calib_write.log 2>&1 && \
tail -1 $O/stats.log && ls $O/*/*/ 
