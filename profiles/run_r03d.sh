mkdir -p gpurun_out/r03d
python -m pytest tests -m gpu -q --deselect "tests/test_gpu_parity.py::test_fit_matches_reference_fixture" > gpurun_out/r03d/gpu_tests_rest.log 2>&1; tail -3 gpurun_out/r03d/gpu_tests_rest.log
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "test_fit_matches_reference_fixture and g12" > gpurun_out/r03d/gpu_tests_g12.log 2>&1; tail -2 gpurun_out/r03d/gpu_tests_g12.log
bash profiles/collect_r03.sh r03d_cfg4 && echo cfg4 done
bash profiles/collect_r03.sh r03d_cfg5s --size cfg5-small && echo cfg5s done
python3 bench.py > gpurun_out/r03d/bench_full.json 2> gpurun_out/r03d/bench_full.err; tail -c 600 gpurun_out/r03d/bench_full.json
python3 profiles/cond_estimates.py > gpurun_out/r03d/cond_estimates.txt 2> gpurun_out/r03d/cond_estimates.err; tail -5 gpurun_out/r03d/cond_estimates.txt
