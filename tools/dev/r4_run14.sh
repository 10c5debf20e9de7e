mkdir -p gpurun_out/r4m
python bench.py > gpurun_out/r4m/bench_c3.json 2> gpurun_out/r4m/bench_c3.err
python bench.py --config c2 > gpurun_out/r4m/bench_c2.json 2> gpurun_out/r4m/bench_c2.err
python bench.py --config c4 --no-secondary > gpurun_out/r4m/bench_c4.json 2> gpurun_out/r4m/bench_c4.err
python bench.py --config c5 --no-secondary --steps 20 > gpurun_out/r4m/bench_c5.json 2> gpurun_out/r4m/bench_c5.err
timeout 600 python -m pytest tests/test_bench_launch.py -m gpu -q -x 2>&1 | tail -3 > gpurun_out/r4m/tests.log
