#!/bin/bash
# Regenerate the artefacts under profiles/ on the GPU box (outputs land in gpurun_out/prof/; copy what is judged
# into profiles/ afterwards).  usage: tools/refresh_profiles.sh [tag]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/bench.py" --no-projection --steps 30 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1     # (a fresh box runs its first process ~2 % slower: not recorded)
python3 "$ROOT/bench.py" 2>/dev/null | tail -1 > "$OUT/bench_line.json"
# strong-scaling operating points of one GPU (per-GPU batch = 256 / N for N = 2, 4, 8): hipGraph-replayed steps
for b in 128 64 32; do python3 "$ROOT/bench.py" --no-projection --batch $b --steps 100 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/bench_line_batch$b.json"; done
# step kinds of configs C3 / C4 (SSL + CM heads active: epoch 5 of DrugLAMP2C2P; SSL epoch of DrugLAMP)
python3 "$ROOT/bench.py" --no-projection --model DrugLAMP2C2P --epoch 5 --steps 50 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/bench_line_2c2p_epoch5.json"
python3 "$ROOT/bench.py" --no-projection --model DrugLAMP --epoch 5 --steps 50 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/bench_line_druglamp_epoch5.json"
# CM steps as hipGraph replays (DrugLAMP2C2P after RS.INIT_EPOCH; epoch 10 is an SSL epoch as well), SSL-epoch step of DrugLAMP, batch 32
python3 "$ROOT/bench.py" --no-projection --model DrugLAMP2C2P --epoch 6 --batch 32 --steps 100 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/bench_line_2c2p_epoch6_cm_batch32.json"
python3 "$ROOT/bench.py" --no-projection --model DrugLAMP2C2P --epoch 10 --batch 32 --steps 100 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/bench_line_2c2p_epoch10_ssl_cm_batch32.json"
python3 "$ROOT/bench.py" --no-projection --model DrugLAMP --epoch 5 --batch 32 --steps 100 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/bench_line_druglamp_epoch5_ssl_batch32.json"
# kernel trace of replayed batch-32 steps (launch count per step = launches / steps in the window)
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace32" -o t -- python3 "$ROOT/bench.py" --no-projection --batch 32 --steps 100 --no-cpu-baseline --no-kernel-timing > "$OUT/b32_under_rocprof.log" 2>&1
T32=$(find "$OUT/trace32" -name '*kernel_trace.csv' | head -1)
python3 "$ROOT/tools/prof_summary.py" "$T32" 0.25 > "$OUT/batch32_graph_kernel_summary.txt" 2>&1
rm -rf "$OUT/trace32"
# the same command with the forward on ONE stream: per-kernel durations that are a kernel's own (with the branches on side
# streams a traced duration includes the kernels it shares the chip with) — what the bench line's roofline object is checked against
export DL_BRANCH_STREAMS=0
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace1" -o t -- python3 "$ROOT/bench.py" --no-projection --steps 10 --warmup 3 > "$OUT/under_rocprof_one_stream.log" 2>&1
unset DL_BRANCH_STREAMS
S1=$(find "$OUT/trace1" -name '*kernel_stats.csv' | head -1); cp "$S1" "$OUT/bench_kernel_stats_one_stream.csv"; rm -rf "$OUT/trace1"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 "$ROOT/bench.py" --no-projection --steps 10 --warmup 3 > "$OUT/under_rocprof.log" 2>&1
grep '"metric"' "$OUT/under_rocprof.log" | tail -1 > "$OUT/bench_line_under_rocprof.json"
T=$(find "$OUT/trace" -name '*kernel_trace.csv' | head -1); S=$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)
cp "$S" "$OUT/bench_kernel_stats.csv"
python3 "$ROOT/tools/prof_summary.py" "$T" 0.25 > "$OUT/bench_timed_window_summary.txt" 2>&1
export DL_BRANCH_STREAMS=0      # (counter collection serialises the kernels anyway; one stream keeps the dispatch order that of the step)
# three passes per counter (two passes of one commit differed by 12 %): the summary reports the median pass and min / max
FS=""; WS=""
for i in 1 2 3; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_f$i" -o f -- python3 "$ROOT/bench.py" --no-projection --steps 2 --warmup 1 --distinct-batches 2 --no-cpu-baseline --no-kernel-timing > "$OUT/pmc_f$i.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_w$i" -o w -- python3 "$ROOT/bench.py" --no-projection --steps 2 --warmup 1 --distinct-batches 2 --no-cpu-baseline --no-kernel-timing > "$OUT/pmc_w$i.log" 2>&1
  FS="$FS,$(find "$OUT/pmc_f$i" -name '*counter_collection.csv' | head -1)"; WS="$WS,$(find "$OUT/pmc_w$i" -name '*counter_collection.csv' | head -1)"
done
unset DL_BRANCH_STREAMS
python3 "$ROOT/tools/pmc_summary.py" --passes "${FS#,}" "${WS#,}" "$OUT/pmc_summary.json" > "$OUT/pmc_summary.txt" 2>&1
python3 "$ROOT/bench.py" --no-projection --graph on --steps 100 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > "$OUT/bench_line_batch256_graph.json"
DL_BRANCH_STREAMS=0 python3 "$ROOT/bench.py" --no-projection --steps 100 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > "$OUT/bench_line_one_stream.json"
DL_CNN_COMPACT=0 python3 "$ROOT/bench.py" --no-projection --steps 100 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > "$OUT/bench_line_cnn_every_position.json"
# round 5 A/B lines (same box, same run): PGCA over all 512 drug rows; ONE static batch instead of eight distinct ones
DL_KEY_COMPACT=0 python3 "$ROOT/bench.py" --no-projection --steps 100 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > "$OUT/bench_line_pgca_all_drug_rows.json"
python3 "$ROOT/bench.py" --no-projection --steps 100 --distinct-batches 1 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > "$OUT/bench_line_one_static_batch.json"
python3 "$ROOT/bench.py" --no-projection --batch 32 --steps 200 --distinct-batches 1 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > "$OUT/bench_line_batch32_one_static_batch.json"
python3 "$ROOT/bench.py" --no-projection --seq-len 9216 --batch 32 --steps 50 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/bench_line_config5_seq9216_batch32.json"
python3 "$ROOT/tools/dgrad_layout_bench.py" --no-projection 2>/dev/null | grep -v amdgpu > "$OUT/dgrad_layout.txt"
python3 "$ROOT/tools/torch_glue_profile.py" 256 1 2>/dev/null | grep -v amdgpu | head -40 > "$OUT/torch_glue_cls_step.txt"
python3 "$ROOT/tools/torch_glue_profile.py" 256 5 2>/dev/null | grep -v amdgpu | head -40 > "$OUT/torch_glue_ssl_step.txt"
python3 "$ROOT/tools/gemm_shapes.py" > "$OUT/gemm_shapes.txt" 2>&1
python3 "$ROOT/tools/gemm_shapes.py" --batch 32 > "$OUT/gemm_shapes_batch32.txt" 2>&1
python3 "$ROOT/tools/cpu_baseline_sweep.py" > "$OUT/cpu_baseline_sweep.jsonl" 2>/dev/null
rm -rf "$OUT/trace" "$OUT"/pmc_f? "$OUT"/pmc_w?
ls -la "$OUT"; head -c 600 "$OUT/bench_line.json"; echo; head -12 "$OUT/bench_timed_window_summary.txt"
