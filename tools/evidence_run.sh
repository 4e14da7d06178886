#!/bin/bash
# Evidence run on a GPU box, from the repo root:  tools/evidence_run.sh r03
# Leaves the bench lines, the rocprofv3 kernel stats of the driver's command, the PMC passes (HBM traffic, MFMA busy, wave
# cycles) and their summaries under gpurun_out/; `python tools/make_profile_summary.py r03` (anywhere) assembles profiles/r03_*.
set -e -o pipefail
TAG=$1
R=$PWD
O=$R/gpurun_out
mkdir -p "$O"
timeout -k 10 300 python bench.py | tail -n 1 > "$O/${TAG}_bench_line.json"
echo headline done
timeout -k 10 300 python bench.py --fps 1 --no-cpu-baseline | tail -n 1 > "$O/${TAG}_bench_fps.json"
timeout -k 10 300 python bench.py --config appearance --D 4 --no-cpu-baseline | tail -n 1 > "$O/${TAG}_bench_appearance.json"
timeout -k 10 400 python bench.py --config dense | tail -n 1 > "$O/${TAG}_bench_dense.json"
echo dense done
timeout -k 10 300 python bench.py --precision x3b --no-cpu-baseline | tail -n 1 > "$O/${TAG}_bench_optin_x3b.json"
timeout -k 10 300 python bench.py --precision x3 --no-cpu-baseline | tail -n 1 > "$O/${TAG}_bench_optin_x3.json"
FACL_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --B 16 --no-cpu-baseline | tail -n 1 > "$O/${TAG}_bench_2rank_gloo_rehearsal.json"
timeout -k 10 300 python bench.py --no-cpu-baseline --rehearse-dp 1 | tail -n 1 > "$O/${TAG}_bench_rehearse_dp.json"
# the opt-in full-graph rehearsal may abort inside torch's process-group watchdog: the evidence run goes on, the line
# of the previous successful run stays
if FACL_DP_GRAPH=full timeout -k 10 300 python bench.py --no-cpu-baseline --rehearse-dp 1 | tail -n 1 > "$O/${TAG}_fullgraph.tmp"; then
  mv "$O/${TAG}_fullgraph.tmp" "$O/${TAG}_bench_rehearse_dp_fullgraph.json"
else
  echo "full-graph rehearsal FAILED (rc=$?)"; rm -f "$O/${TAG}_fullgraph.tmp"
fi
timeout -k 10 300 python bench.py --config extract | tail -n 1 > "$O/${TAG}_bench_extract.json"
FACL_EVAL_FUSED=0 timeout -k 10 300 python bench.py --config extract --no-cpu-baseline | tail -n 1 > "$O/${TAG}_bench_extract_unfused.json"
timeout -k 10 300 python tools/time_views.py | tail -n 1 > "$O/${TAG}_views.json"
echo bench lines done
# `lines` as second argument: only refresh the bench lines (e.g. after profiles/pmc_*.json were regenerated from this build)
if [ "$2" = "lines" ]; then exit 0; fi
cd /tmp
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$TAG" -o r -- python3 "$R/bench.py" > "$O/prof_${TAG}_bench.log" 2>&1
echo kernel stats done
# PMC passes: counters only with --kernel-trace (separate runs; never with other trace domains)
P="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --graph 0"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/pmc_${TAG}_fetch" -o f -- $P > "$O/pmc_${TAG}_fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/pmc_${TAG}_write" -o w -- $P > "$O/pmc_${TAG}_write.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d "$O/pmc_${TAG}_a" -o a -- $P > "$O/pmc_${TAG}_a.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d "$O/pmc_${TAG}_b" -o b -- $P > "$O/pmc_${TAG}_b.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d "$O/pmc_${TAG}_c" -o c -- $P > "$O/pmc_${TAG}_c.log" 2>&1
cd "$R"
python tools/prof_summary.py "$O/prof_$TAG" 56 44 > "$O/prof_${TAG}_summary.txt"
python tools/pmc_summary.py "$O/pmc_${TAG}_fetch" "$O/pmc_${TAG}_write" 24 > "$O/pmc_${TAG}_traffic.md"
python tools/pmc_mfma_summary.py "$O/pmc_${TAG}_a" "$O/pmc_${TAG}_b" "$O/pmc_${TAG}_mfma_util.json" > "$O/pmc_${TAG}_mfma_util.md"
python tools/pmc_wave_summary.py "$O/pmc_${TAG}_a" "$O/pmc_${TAG}_c" > "$O/pmc_${TAG}_wave_cycles.md"
# keep the merged-back payload small: the raw counter CSVs are not needed once summarised
rm -rf "$O/pmc_${TAG}_fetch" "$O/pmc_${TAG}_write" "$O/pmc_${TAG}_a" "$O/pmc_${TAG}_b" "$O/pmc_${TAG}_c"
find "$O/prof_$TAG" -name "*kernel_trace.csv" -delete
echo evidence run "$TAG" done
