#!/bin/bash
# usage: pmc_quick.sh TAG [bench args]   (env FA_LIB / FA_PATCH_LDS are inherited)
export TMPDIR=/tmp; R=$PWD; tag=$1; shift
run() { name=$1; shift; timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_${tag}_$name -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-elems 0 $BARGS > gpurun_out/pmc_${tag}_$name.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/pmc_${tag}_$name.log; return 1; }; }
BARGS="$*"
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS && \
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS && \
timeout -s KILL 240 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt_${tag} -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-elems 0 $BARGS > gpurun_out/kt_${tag}.log 2>&1
python3 scripts/pmc_summary.py gpurun_out/pmc_${tag}_sq1 gpurun_out/pmc_${tag}_sq2
grep assemble gpurun_out/kt_${tag}/*/*kernel_trace.csv | head -2 | cut -d, -f8-
