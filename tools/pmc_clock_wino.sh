#!/bin/bash
# Clock / matrix-pipe occupancy / wait shares of the Winograd-path kernels at per-GPU batch $1 -> gpurun_out/probe_clock_wino_bs$1.txt
# (two PMC passes, each its own rocprofv3 run with --kernel-trace only).  Usage: bash tools/pmc_clock_wino.sh 128
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
B=$1
O=$R/gpurun_out/pmc_clock_wino_bs$B
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O -o a -- python3 $R/tools/probe_clock.py run-wino $B > $O/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O -o b -- python3 $R/tools/probe_clock.py run-wino $B > $O/b.log 2>&1
{
  echo "# tools/pmc_clock_wino.sh $B: fp32 default routing (F(4x4,3x3) Winograd), layers $(python3 -c "import sys; sys.path.insert(0,'$R/tools'); import probe_clock as p; print(p.WINO_LAYERS)") at B = $B"
  echo "# pass 1: GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES (median of the last 3 dispatches per kernel and grid)"
  python3 $R/tools/probe_clock.py report $(find $O -name 'a_counter_collection.csv' | head -1)
  echo "# pass 2: SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
  python3 $R/tools/probe_clock.py report $(find $O -name 'b_counter_collection.csv' | head -1)
} > $R/gpurun_out/probe_clock_wino_bs$B.txt
find $O -name '*counter_collection.csv' -delete
find $O -name '*kernel_trace.csv' -delete
