#!/bin/bash
# One-call evidence refresh on the GPU box:  bash profiles/collect_r02.sh <tag>
# bench line (with CPU baseline), rocprofv3 kernel stats of the same command, PMC passes, the other sizes.
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py --steps 5 --warmup 2 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG} -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline \
    > gpurun_out/${TAG}_bench_profiled.json 2> gpurun_out/${TAG}_bench_profiled.err
bash profiles/collect_pmc.sh ${TAG}
python3 profiles/pmc_to_json.py ${TAG} gpurun_out/${TAG}_pmc_k_row_tasks.json
for s in cfg2 cfg3 cfg5-small; do
  python3 bench.py --size $s --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_size_$s.json 2> gpurun_out/${TAG}_size_$s.err
done
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --solve-dtype float64 > gpurun_out/${TAG}_size_cfg4-f64.json 2> /dev/null
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --graph product > gpurun_out/${TAG}_size_cfg4-product-graph.json 2> /dev/null
python3 bench.py --size cfg5 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_size_cfg5.json 2> gpurun_out/${TAG}_size_cfg5.err
echo collected
