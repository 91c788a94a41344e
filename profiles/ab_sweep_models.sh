# placed dataflow sweep: timing-model sweep on the GPU box (bench.py cfg4, 5 steps) - which `pre` matches the hardware?
mkdir -p gpurun_out/r03g
run() {  # label env...
  label=$1; shift
  env "$@" timeout -k 10 120 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/r03g/q.json 2> gpurun_out/r03g/q.err
  python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); p=d['phase_ms_per_step']
print('%-44s iter %.3f gs %.3f rmse %.12f' % (sys.argv[2], d['ms_per_step'], p['gs_sweep'], d['train_rmse'][-1]), flush=True)" gpurun_out/r03g/q.json "$label"
}
run "round-robin" ALS_GS_PLACED=0
for pre in 3 6 10 15 25; do
  run "placed nomail pre=$pre" ALS_GS_NOMAIL=1 ALS_GS_MODEL=$pre,1.2,3.5,0.3,0.5
  run "placed mail   pre=$pre" ALS_GS_NOMAIL=0 ALS_GS_MODEL=$pre,1.2,3.5,0.3,0.5
done
