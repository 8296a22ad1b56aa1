#!/bin/bash
# counters of the cfg4 tile kernel (Hex8 phase field, 100^3)
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r02_cfg4; mkdir -p $O
run() { name=$1; shift; (cd /tmp && timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$name -- python3 $R/bench.py --workload cfg4 --n 100 --steps 2 --warmup 1 --cpu-elems 0 --no-f-int-only > $O/pmc_$name.log 2>&1) || { echo "pass $name failed"; tail -5 $O/pmc_$name.log; return 1; }; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS && \
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS && \
run sq3 SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_WAVE32_LDS SQ_LDS_ADDR_CONFLICT
python3 scripts/pmc_summary.py $O/pmc_sq1 $O/pmc_sq2 $O/pmc_sq3 > $O/pmc_summary.txt; cat $O/pmc_summary.txt
