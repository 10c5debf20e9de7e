set -x
mkdir -p gpurun_out/r4d
python -m pytest tests -m gpu -q -x 2>&1 | tail -40 > gpurun_out/r4d/tests.log
python -m pytest tests/test_gpu_multistart.py -m gpu -q -s 2>&1 | grep -v "^$" | tail -80 > gpurun_out/r4d/multistart.log
python bench.py > gpurun_out/r4d/bench.json 2> gpurun_out/r4d/bench.err
