export TMPDIR=/tmp
rm -rf gpurun_out/pmc_pair
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_pair -- python3 bench.py --precision f16 --batch 8 --height 2160 --width 3840 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_pair.log 2>&1
rm -rf gpurun_out/pmc_pair2
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_pair2 -- python3 bench.py --precision f16 --batch 8 --height 2160 --width 3840 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_pair2.log 2>&1
tail -2 gpurun_out/pmc_pair2.log
