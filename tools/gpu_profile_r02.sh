#!/bin/bash
# Round-2 rocprofv3 evidence run (on the GPU box, from the repo root): kernel trace + stats of bench.py, per-step breakdown,
# and three separate --pmc passes over tools/pmc_conv_pass.py (MFMA busy cycles, FETCH_SIZE, WRITE_SIZE).
set -x
R=$(pwd)
O=$R/gpurun_out/r02h
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -o bench -- python "$R/bench.py" --steps 24 --warmup 8 --no-kernel-rooflines --no-cpu-baseline --no-input-pipeline > "$O/bench_trace.json" 2> "$O/bench_trace.err"
KT=$(find "$O/trace" -name "*kernel_trace.csv" | head -1)
python "$R/tools/step_breakdown.py" "$KT" > "$O/train_step_breakdown.txt"
head -30 "$O/train_step_breakdown.txt"
KS=$(find "$O/trace" -name "*kernel_stats.csv" | head -1)
cp "$KS" "$O/bench_kernel_stats.csv"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/pmc_mfma" -o p -- python "$R/tools/pmc_conv_pass.py" 256 > "$O/pmc_mfma.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch" -o FETCH_SIZE -- python "$R/tools/pmc_conv_pass.py" 256 > "$O/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write" -o WRITE_SIZE -- python "$R/tools/pmc_conv_pass.py" 256 > "$O/pmc_write.log" 2>&1
find "$O" -name "*counter_collection.csv"
find "$O" -name "*.db" -delete
find "$O" -name "*kernel_trace.csv" -size +30M -delete
du -sh "$O"
