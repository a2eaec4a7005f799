#!/bin/bash
# Round-6 rocprofv3 evidence run (GPU box, repo root): kernel trace + stats of bench.py's train loop, per-step breakdown, and three separate
# --pmc passes over tools/pmc_conv_pass.py (MFMA busy cycles; FETCH_SIZE; WRITE_SIZE — counters only next to --kernel-trace).
R=$(pwd); O=$R/gpurun_out/${1:-r06prof}; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -o bench -- python "$R/bench.py" --steps 24 --warmup 8 --no-kernel-rooflines --no-cpu-baseline > "$O/bench_under_rocprof.json" 2> "$O/bench_trace.err"
KT=$(find "$O/trace" -name "*kernel_trace.csv" | head -1)
python "$R/tools/step_breakdown.py" "$KT" > "$O/train_step_breakdown.txt"
KS=$(find "$O/trace" -name "*kernel_stats.csv" | head -1); cp "$KS" "$O/bench_kernel_stats.csv"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/pmc_mfma" -o p -- python "$R/tools/pmc_conv_pass.py" 256 > "$O/pmc_mfma.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch" -o FETCH_SIZE -- python "$R/tools/pmc_conv_pass.py" 256 > "$O/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write" -o WRITE_SIZE -- python "$R/tools/pmc_conv_pass.py" 256 > "$O/pmc_write.log" 2>&1
mkdir -p "$O/cc"; find "$O" -name "*counter_collection.csv" -exec cp {} "$O/cc/" \;
cd "$R"
python tools/pmc_mfma_parse.py "$O/cc/p_counter_collection.csv" r06_conv_mfma_util.json | tail -60 > "$O/mfma_util.txt"
python tools/pmc_conv_parse.py "$O/cc" r06_conv_pmc_traffic.json | tail -14 > "$O/pmc_traffic.txt"
cp profiles/r06_conv_mfma_util.json profiles/r06_conv_pmc_traffic.json "$O/"
find "$O" -name "*.db" -delete; find "$O" -name "*kernel_trace.csv" -delete; find "$O" -name "*counter_collection.csv" -size +20M -delete
head -40 "$O/train_step_breakdown.txt"; head -8 "$O/mfma_util.txt"; du -sh "$O"
