mkdir -p gpurun_out/r03w
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -x -q -k "graph or sweep or gs_ or k128 or g10_ or g5_ or reproducible or cfg5 or stream" > gpurun_out/r03w/tests.log 2>&1; tail -3 gpurun_out/r03w/tests.log
one() {  # label size env...
  label=$1; size=$2; shift 2
  env "$@" timeout -k 10 300 python3 bench.py --size $size --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/r03w/q.json 2> gpurun_out/r03w/q.err
  python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); p=d['phase_ms_per_step']
print('%-10s %-10s iter %.3f user %.3f item %.3f gs %.3f rmse %.12f' % (sys.argv[2], sys.argv[3], d['ms_per_step'], p.get('row_solve_user',0), p.get('row_solve_item',0), p.get('gs_sweep',0), d['train_rmse'][-1]), flush=True)" gpurun_out/r03w/q.json "$label" $size || tail -5 gpurun_out/r03w/q.err
}
PREV=$PWD/collaborative-filtering_amd/csrc/libals_hip_prev.so
for rep in 1 2; do
  one prev cfg5-small ALS_HIP_LIB=$PREV
  one new cfg5-small X=1
done
for size in k80 k96 k160; do
  one prev $size ALS_HIP_LIB=$PREV
  one new $size X=1
done
