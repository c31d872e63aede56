#!/bin/bash
# Round profile pass on the GPU box: kernel stats of the bench command, the two PMC passes for the HBM-side
# traffic of sg_igemm_kernel, and a small-batch kernel trace.  Usage: bash tools/gpu_profile.sh OUTDIR
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/${1:-gpurun_out/prof}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --no-cpu-baseline --no-kernel-timing"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats128 -o run -- python3 $B --steps 2 --warmup 1 > $O/stats128.log 2>&1
echo stats128 done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats16 -o run -- python3 $B --steps 4 --warmup 2 --batch 16 > $O/stats16.log 2>&1
echo stats16 done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o run -- python3 $B --steps 1 --warmup 1 > $O/pmc_fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o run -- python3 $B --steps 1 --warmup 1 > $O/pmc_write.log 2>&1
echo write done
python3 $R/tools/pmc_traffic.py $O/pmc_fetch/run_counter_collection.csv $O/pmc_write/run_counter_collection.csv --kernel sg_igemm_kernel \
  --command "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- python3 bench.py --no-cpu-baseline --no-kernel-timing --steps 1 --warmup 1" \
  --out $O/igemm_traffic_bs128.json > $O/traffic.log 2>&1
python3 $R/tools/pmc_traffic.py $O/pmc_fetch/run_counter_collection.csv $O/pmc_write/run_counter_collection.csv --kernel sg_wgrad_kernel \
  --out $O/wgrad_traffic_bs128.json >> $O/traffic.log 2>&1
# the raw per-dispatch counter files are large: keep only the summaries
rm -f $O/pmc_fetch/run_counter_collection.csv $O/pmc_write/run_counter_collection.csv $O/*/run_kernel_trace.csv.bak
echo profile pass done
