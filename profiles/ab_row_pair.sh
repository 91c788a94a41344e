#!/bin/bash
# A/B of the two-waves-per-row kernel (row_pair.hip) against one wave per row at k = 128:
#   bash profiles/ab_row_pair.sh <tag> [size]      (size: cfg5-small (default) or cfg5)
TAG=$1
SIZE=${2:-cfg5-small}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in 1 0; do
  export ALS_ROW_PAIR=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_pair$v -- python3 bench.py --size $SIZE --steps 5 --warmup 2 --no-cpu-baseline \
      > gpurun_out/${TAG}_pair${v}_bench.json 2> gpurun_out/${TAG}_pair${v}_bench.err || exit 1
  python3 profiles/kernel_stats_md.py gpurun_out/prof_${TAG}_pair$v gpurun_out/${TAG}_pair${v}_kernel_stats.md "${TAG} ALS_ROW_PAIR=$v" --size $SIZE --steps 5 --warmup 2 --no-cpu-baseline
done
echo done
