#!/bin/bash
# SQ counter passes over tools/bh_prof.py (argument: N, tag); one rocprofv3 run per counter set.
N=${1:-1048576}; TAG=${2:-new}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name counters...
  name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout -s KILL 120 rocprofv3 --pmc "$@" -d /tmp/pmc_$name --output-format csv -- python3 $R/tools/bh_prof.py $N > /tmp/pmc_$name.log 2>&1 || { echo "pass $name failed"; tail -5 /tmp/pmc_$name.log; return 1; }
  python3 $R/tools/pmc_sum.py /tmp/pmc_$name > $R/gpurun_out/pmc_${TAG}_${N}_$name.json
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY && \
run b SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVES && \
run c GRBM_GUI_ACTIVE
