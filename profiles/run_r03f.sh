# other shapes with the final build: cfg 2 (sustained over 100 iterations) and cfg 5 at full size on one GPU
mkdir -p gpurun_out/r03f
python3 bench.py --size cfg2 --steps 100 --warmup 5 --no-cpu-baseline > gpurun_out/r03f/cfg2.json 2> gpurun_out/r03f/cfg2.err; tail -c 400 gpurun_out/r03f/cfg2.json; echo
python3 bench.py --size cfg5 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03f/cfg5.json 2> gpurun_out/r03f/cfg5.err; tail -c 700 gpurun_out/r03f/cfg5.json; echo
