set -x
mkdir -p gpurun_out/r6a
export DL_PARITY_LOG=$PWD/gpurun_out/r6a/margins.jsonl
rm -f $DL_PARITY_LOG
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_northstar_block_gpu.py "tests/test_kernels_gpu.py::test_cross_entropy_rows_against_torch" tests/test_model_gpu.py tests/test_losses_gpu.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r6a/pytest_parity.txt
unset DL_PARITY_LOG
python tools/parity_margins.py gpurun_out/r6a/margins.jsonl > gpurun_out/r6a/parity_margins.txt 2>&1
timeout 300 python tools/northstar_block.py > gpurun_out/r6a/northstar_block.txt 2>&1
timeout 600 python bench.py --steps 100 > gpurun_out/r6a/bench_line.json 2> gpurun_out/r6a/bench_err.txt
tail -c 1500 gpurun_out/r6a/pytest_parity.txt; cat gpurun_out/r6a/northstar_block.txt; head -c 900 gpurun_out/r6a/bench_line.json
