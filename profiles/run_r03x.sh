mkdir -p gpurun_out/r03x
ALS_GS_FORM=stream python3 profiles/debug/sweep_form_probe.py 80 100 112 128 144 150 2>/dev/null
ALS_HIP_LIB=$PWD/collaborative-filtering_amd/csrc/libals_hip_prev.so ALS_GS_FORM=stream python3 profiles/debug/sweep_form_probe.py 80 100 112 128 144 150 2>/dev/null
one() {  # label size env...
  label=$1; size=$2; shift 2
  env "$@" timeout -k 10 300 python3 bench.py --size $size --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/r03x/q.json 2> gpurun_out/r03x/q.err
  python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); p=d['phase_ms_per_step']
print('%-14s %-10s iter %.3f user %.3f item %.3f gs %.3f rmse %.12f' % (sys.argv[2], sys.argv[3], d['ms_per_step'], p.get('row_solve_user',0), p.get('row_solve_item',0), p.get('gs_sweep',0), d['train_rmse'][-1]), flush=True)" gpurun_out/r03x/q.json "$label" $size || tail -5 gpurun_out/r03x/q.err
}
PREV=$PWD/collaborative-filtering_amd/csrc/libals_hip_prev.so
one "prev stream" cfg5-small ALS_HIP_LIB=$PREV ALS_GS_FORM=stream
one "new stream" cfg5-small ALS_GS_FORM=stream
one "prev stream" k96 ALS_HIP_LIB=$PREV ALS_GS_FORM=stream
one "new stream" k96 ALS_GS_FORM=stream
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "stream or other_k or g10_full_k128" 2>&1 | tail -2
