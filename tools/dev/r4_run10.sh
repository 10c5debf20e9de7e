mkdir -p gpurun_out/r4j
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > gpurun_out/r4j/tests.log
python tools/fit_only.py c3 > gpurun_out/r4j/fit_c3.txt 2>&1
python tools/fit_only.py c2 > gpurun_out/r4j/fit_c2.txt 2>&1
python tools/fit_only.py c4 > gpurun_out/r4j/fit_c4.txt 2>&1
python bench.py --no-cpu-baseline > gpurun_out/r4j/bench.json 2> gpurun_out/r4j/bench.err
