export TMPDIR=/tmp; R=$PWD
run() { name=$1; shift; timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$name -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-elems 0 > gpurun_out/pmc_$name.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/pmc_$name.log; return 1; }; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS && \
run fetch FETCH_SIZE && \
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum && \
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
ls gpurun_out/pmc_*/*/ | head -30
