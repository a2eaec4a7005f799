#!/bin/bash
# SQ wave-state counters of the conv kernels (one launch pair per ResNet-50 conv configuration): where do the waves wait?
R=$(pwd); O=$R/gpurun_out/r02j; mkdir -p "$O"; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d "$O/pmc_sq" -o sq -- python "$R/tools/pmc_conv_pass.py" 256 > "$O/pmc_sq.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d "$O/pmc_sq2" -o sq2 -- python "$R/tools/pmc_conv_pass.py" 256 > "$O/pmc_sq2.log" 2>&1
find "$O" -name "*.db" -delete
ls "$O"/*
