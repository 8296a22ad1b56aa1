#!/bin/bash
# Hex20: MFMA-FP64 experiment + counters of the current cfg5 kernel (64^3 block, Gauss3^3)
export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out/r02_hex20
timeout -k 5 120 ./scripts/calib/hex20_mfma 32768 > gpurun_out/r02_hex20/mfma_experiment.txt 2>&1 || { echo "mfma experiment failed"; tail -5 gpurun_out/r02_hex20/mfma_experiment.txt; exit 1; }
cat gpurun_out/r02_hex20/mfma_experiment.txt
run() { name=$1; shift; (cd /tmp && timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/r02_hex20/pmc_$name -- python3 $R/bench.py --workload cfg5 --n 64 --steps 2 --warmup 1 --cpu-elems 0 > $R/gpurun_out/r02_hex20/pmc_$name.log 2>&1) || { echo "pass $name failed"; tail -5 gpurun_out/r02_hex20/pmc_$name.log; return 1; }; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS && \
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
python3 scripts/pmc_summary.py gpurun_out/r02_hex20/pmc_sq1 gpurun_out/r02_hex20/pmc_sq2 > gpurun_out/r02_hex20/pmc_summary.txt; cat gpurun_out/r02_hex20/pmc_summary.txt
timeout -k 10 200 python bench.py --workload cfg5 --n 64 --steps 5 --warmup 2 --cpu-elems 0 2>/dev/null | tail -1 > gpurun_out/r02_hex20/cfg5_n64.json; cat gpurun_out/r02_hex20/cfg5_n64.json | cut -c1-400
