#!/bin/bash
# counters of the Hex20 MFMA-FP64 experiment (scripts/calib/hex20_mfma): one pass, microbenchmark binary directly after --
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r02_hex20; mkdir -p $O
timeout -k 5 120 ./scripts/calib/hex20_mfma 32768 > $O/mfma_experiment.txt 2>&1 || { echo "experiment failed"; tail -5 $O/mfma_experiment.txt; exit 1; }
cat $O/mfma_experiment.txt
(cd /tmp && timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_mfma -- $R/scripts/calib/hex20_mfma 32768 > $O/pmc_mfma.log 2>&1) || { echo "pmc pass failed"; tail -5 $O/pmc_mfma.log; exit 1; }
python3 scripts/pmc_summary.py $O/pmc_mfma > $O/pmc_mfma_summary.txt && cat $O/pmc_mfma_summary.txt
