#!/bin/bash
# Three separate --pmc passes over tools/pmc_conv_pass.py (every conv configuration of ResNet-50 at batch 256, product heuristic):
# MFMA busy cycles, FETCH_SIZE, WRITE_SIZE -> profiles/r02_conv_mfma_util.json, profiles/r02_conv_pmc_traffic.json
R=$(pwd); O=$R/gpurun_out/${1:-pmc_conv}; mkdir -p "$O"; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/pmc_mfma" -o p -- python "$R/tools/pmc_conv_pass.py" 256 > "$O/pmc_mfma.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch" -o FETCH_SIZE -- python "$R/tools/pmc_conv_pass.py" 256 > "$O/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write" -o WRITE_SIZE -- python "$R/tools/pmc_conv_pass.py" 256 > "$O/pmc_write.log" 2>&1
mkdir -p "$O/cc"; find "$O" -name "*counter_collection.csv" -exec cp {} "$O/cc/" \;
cd "$R"
python tools/pmc_mfma_parse.py "$O/cc/p_counter_collection.csv" | tail -12
python tools/pmc_conv_parse.py "$O/cc" | tail -14
cp profiles/r02_conv_mfma_util.json profiles/r02_conv_pmc_traffic.json "$O/"
find "$O" -name "*.db" -delete; find "$O" -name "*kernel_trace.csv" -delete
