mkdir -p gpurun_out/r03u
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q > gpurun_out/r03u/kernel_tests.log 2>&1; tail -3 gpurun_out/r03u/kernel_tests.log
one() {  # label size env...
  label=$1; size=$2; shift 2
  env "$@" timeout -k 10 300 python3 bench.py --size $size --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/r03u/q.json 2> gpurun_out/r03u/q.err
  python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); p=d['phase_ms_per_step']
print('%-22s %-10s iter %.3f user %.3f item %.3f gs %.3f rmse %.12f' % (sys.argv[2], sys.argv[3], d['ms_per_step'], p.get('row_solve_user',0), p.get('row_solve_item',0), p.get('gs_sweep',0), d['train_rmse'][-1]), flush=True)" gpurun_out/r03u/q.json "$label" $size || tail -5 gpurun_out/r03u/q.err
}
for rep in 1 2; do
  one "in-kernel split" cfg5-small ALS_PLANES_K128=0
  one "pre-split planes" cfg5-small ALS_PLANES_K128=1
done
