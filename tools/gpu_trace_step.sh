#!/bin/bash
# rocprofv3 kernel trace + stats of bench.py (train loop, no extra probes) and the per-train-step kernel breakdown.
# usage (GPU box, repo root): bash tools/gpu_trace_step.sh <out-dir-under-gpurun_out>
set -x
R=$(pwd)
O=$R/gpurun_out/${1:-trace}
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -o bench -- python "$R/bench.py" --steps 24 --warmup 8 --no-kernel-rooflines --no-cpu-baseline > "$O/bench_under_rocprof.json" 2> "$O/bench_trace.err"
KT=$(find "$O/trace" -name "*kernel_trace.csv" | head -1)
python "$R/tools/step_breakdown.py" "$KT" > "$O/train_step_breakdown.txt"
KS=$(find "$O/trace" -name "*kernel_stats.csv" | head -1)
cp "$KS" "$O/bench_kernel_stats.csv"
find "$O" -name "*.db" -delete
find "$O/trace" -name "*kernel_trace.csv" -delete
head -45 "$O/train_step_breakdown.txt"
