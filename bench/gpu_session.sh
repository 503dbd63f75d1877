set -o pipefail
mkdir -p gpurun_out/r2
timeout -k 10 900 python -u -m pytest tests/test_xgmi_gpu.py tests/test_session_gpu.py tests/test_bench_contract.py -m gpu -x -q --timeout 400 --timeout-method thread 2>&1 | tail -25
