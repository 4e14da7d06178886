#!/bin/bash
# Evidence run on a GPU box, from the repo root:  tools/evidence_run.sh r02
# Leaves the four bench lines, the rocprofv3 kernel stats of the driver's command and their summary under gpurun_out/;
# `python tools/make_profile_summary.py r02` (anywhere) then assembles profiles/r02_*.
set -e -o pipefail
TAG=$1
R=$PWD
O=$R/gpurun_out
mkdir -p "$O"
timeout -k 10 300 python bench.py | tail -n 1 > "$O/${TAG}_bench_line.json"
timeout -k 10 300 python bench.py --config appearance --D 4 --no-cpu-baseline | tail -n 1 > "$O/${TAG}_bench_appearance.json"
timeout -k 10 300 python bench.py --config dense --no-cpu-baseline | tail -n 1 > "$O/${TAG}_bench_dense.json"
timeout -k 10 300 python bench.py --precision x3b --no-cpu-baseline | tail -n 1 > "$O/${TAG}_bench_optin_x3b.json"
timeout -k 10 300 python bench.py --precision x3 --no-cpu-baseline | tail -n 1 > "$O/${TAG}_bench_optin_x3.json"
FACL_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --B 16 --no-cpu-baseline | tail -n 1 > "$O/${TAG}_bench_2rank_gloo_rehearsal.json"
cd /tmp
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$TAG" -o r -- python3 "$R/bench.py" > "$O/prof_${TAG}_bench.log" 2>&1
cd "$R"
python tools/prof_summary.py "$O/prof_$TAG" 34 40 > "$O/prof_${TAG}_summary.txt"
echo evidence run "$TAG" done
