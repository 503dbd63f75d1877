set -o pipefail
mkdir -p gpurun_out/r2
timeout -k 10 900 python -u -m pytest tests/test_convergence_gpu.py -m gpu -x -q --timeout 600 --timeout-method thread 2>&1 | tail -25
