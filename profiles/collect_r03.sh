#!/bin/bash
# rocprofv3 kernel-trace summary of a bench.py command on the box:  collect_r03.sh <tag> [bench args ...]
# -> gpurun_out/<tag>_kernel_stats.md + gpurun_out/<tag>_bench_profiled.json   (copy what is to be judged into profiles/)
TAG=$1; shift
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary "$@" \
    > gpurun_out/${TAG}_bench_profiled.json 2> gpurun_out/${TAG}_bench_profiled.err
python3 profiles/kernel_stats_md.py /tmp/prof_$TAG gpurun_out/${TAG}_kernel_stats.md "$TAG" "$@"
