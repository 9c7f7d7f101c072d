#!/bin/bash
# The rocprofv3 runs behind profiles/ (argument: a tag such as r02).  Each counter set is a run of
# its own; nothing else is traced in a counter run.  Results land in gpurun_out/prof_<tag>/.
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
set -e
# 1. kernel trace + stats of the default bench command, and the line it printed
rm -rf /tmp/kt
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d /tmp/kt --output-format csv -- python3 $R/bench.py > $O/bench_line.json 2> $O/bench.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
python3 $R/tools/trace_stats.py /tmp/kt > $O/bench_kernels_by_grid.json
# 2. HBM traffic of the all-pairs kernels: FETCH_SIZE and WRITE_SIZE in separate passes
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout -s KILL 300 rocprofv3 --pmc $c -d /tmp/pmc_$c --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-extras > /dev/null 2> $O/pmc_$c.err
  python3 $R/tools/pmc_sum.py /tmp/pmc_$c > $O/pmc_allpairs_$c.json
done
# 3. HBM traffic of the octree build and the traversal
for n in 1048576 16777216; do for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_bh
  timeout -s KILL 300 rocprofv3 --pmc $c -d /tmp/pmc_bh --output-format csv -- python3 $R/tools/bh_prof.py $n > /dev/null 2> $O/pmc_bh.err
  python3 $R/tools/pmc_sum.py /tmp/pmc_bh > $O/pmc_bh_${n}_$c.json
  # ... and with the reference's own law (law id 1) in the traversal
  rm -rf /tmp/pmc_bh
  timeout -s KILL 300 rocprofv3 --pmc $c -d /tmp/pmc_bh --output-format csv -- python3 $R/tools/bh_prof.py $n 1 > /dev/null 2> $O/pmc_bh.err
  python3 $R/tools/pmc_sum.py /tmp/pmc_bh > $O/pmc_bh_${n}_wgsl_bias_$c.json
done; done
echo done
