#!/bin/bash
# Round profile pass on the GPU box: kernel stats of the bench command (fp32 bs 128 = the headline, bs 16 = the 8-way
# shard size, bf16 bs 256 = config c3), the separate PMC passes for the HBM-side traffic of the dominant conv kernels, and
# SQ counters of the bf16 kernels on the 1024->1024 layer.  Usage: bash tools/gpu_profile.sh OUTDIR   (ONLY_F32=1: the fp32 passes alone)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/${1:-gpurun_out/prof}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --no-cpu-baseline --no-kernel-timing --no-extra-configs"
# (round 4) the production path runs S's passes on a second stream: a kernel's duration in THAT trace includes the time it shares the chip.
# stats128 = one stream (SG_NET_STREAM=0): each kernel has the GPU to itself -- the durations bench.py's roofline is computed from;
# stats128_streams = the default two-stream schedule (what the timed region runs).
SG_NET_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats128 -o run -- python3 $B --steps 2 --warmup 1 > $O/stats128.log 2>&1
echo stats128 done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats128_streams -o run -- python3 $B --steps 2 --warmup 1 > $O/stats128_streams.log 2>&1
echo stats128_streams done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats16 -o run -- python3 $B --steps 4 --warmup 2 --batch 16 > $O/stats16.log 2>&1
echo stats16 done
if [ -z "$ONLY_F32" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bf16 -o run -- python3 $B --steps 2 --warmup 1 --conv-dtype bf16 --batch 256 > $O/stats_bf16.log 2>&1
echo stats_bf16 done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fp8 -o run -- python3 $B --steps 2 --warmup 1 --conv-dtype fp8 --batch 512 --balance > $O/stats_fp8.log 2>&1
echo stats_fp8 done
fi
export SG_NET_STREAM=0      # counter passes: one stream (the counters are per dispatch; overlapped dispatches would share them)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o run -- python3 $B --steps 1 --warmup 1 > $O/pmc_fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o run -- python3 $B --steps 1 --warmup 1 > $O/pmc_write.log 2>&1
echo write done
CMD="rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- python3 bench.py --no-cpu-baseline --no-kernel-timing --steps 1 --warmup 1"
python3 $R/tools/pmc_traffic.py $O/pmc_fetch/run_counter_collection.csv $O/pmc_write/run_counter_collection.csv --kernel sg_igemm \
  --command "$CMD" --out $O/igemm_traffic_bs128.json > $O/traffic.log 2>&1
python3 $R/tools/pmc_traffic.py $O/pmc_fetch/run_counter_collection.csv $O/pmc_write/run_counter_collection.csv --kernel sg_wgrad_kernel \
  --command "$CMD" --out $O/wgrad_traffic_bs128.json >> $O/traffic.log 2>&1
python3 $R/tools/pmc_traffic.py $O/pmc_fetch/run_counter_collection.csv $O/pmc_write/run_counter_collection.csv --kernel "sg_igemm_bf16v2_kernel<128, 4, false, 128" \
  --command "$CMD" --out $O/igemm_wino_traffic_bs128.json >> $O/traffic.log 2>&1
python3 $R/tools/pmc_traffic.py $O/pmc_fetch/run_counter_collection.csv $O/pmc_write/run_counter_collection.csv --kernel "sg_wgrad_kernel<128, 128, 2, 2, 16, 3, true, 0>" \
  --command "$CMD" --out $O/wgrad_wino_traffic_bs128.json >> $O/traffic.log 2>&1
python3 $R/tools/pmc_traffic.py $O/pmc_fetch/run_counter_collection.csv $O/pmc_write/run_counter_collection.csv --kernel k_w \
  --command "$CMD" --out $O/wino_transform_traffic_bs128.json >> $O/traffic.log 2>&1
rm -f $O/pmc_fetch/run_counter_collection.csv $O/pmc_write/run_counter_collection.csv
if [ -n "$ONLY_F32" ]; then echo profile pass done; exit 0; fi
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch16 -o run -- python3 $B --steps 1 --warmup 1 --conv-dtype bf16 --batch 256 > $O/pmc_fetch16.log 2>&1
echo fetch bf16 done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write16 -o run -- python3 $B --steps 1 --warmup 1 --conv-dtype bf16 --batch 256 > $O/pmc_write16.log 2>&1
echo write bf16 done
python3 $R/tools/pmc_traffic.py $O/pmc_fetch16/run_counter_collection.csv $O/pmc_write16/run_counter_collection.csv --kernel sg_igemm_bf16v2_kernel \
  --command "$CMD --conv-dtype bf16 --batch 256" --out $O/igemm_bf16_traffic_bs256.json >> $O/traffic.log 2>&1
python3 $R/tools/pmc_traffic.py $O/pmc_fetch16/run_counter_collection.csv $O/pmc_write16/run_counter_collection.csv --kernel sg_wgrad_bf16v2_kernel \
  --command "$CMD --conv-dtype bf16 --batch 256" --out $O/wgrad_bf16_traffic_bs256.json >> $O/traffic.log 2>&1
rm -f $O/pmc_fetch16/run_counter_collection.csv $O/pmc_write16/run_counter_collection.csv $O/*/run_kernel_trace.csv.bak
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch8 -o run -- python3 $B --steps 1 --warmup 1 --conv-dtype fp8 --batch 512 --balance > $O/pmc_fetch8.log 2>&1
echo fetch fp8 done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write8 -o run -- python3 $B --steps 1 --warmup 1 --conv-dtype fp8 --batch 512 --balance > $O/pmc_write8.log 2>&1
echo write fp8 done
python3 $R/tools/pmc_traffic.py $O/pmc_fetch8/run_counter_collection.csv $O/pmc_write8/run_counter_collection.csv --kernel "sg_igemm_bf16v2_kernel<256, 1" \
  --command "$CMD --conv-dtype fp8 --batch 512 --balance" --out $O/igemm_fp8_traffic_bs512.json >> $O/traffic.log 2>&1
python3 $R/tools/pmc_traffic.py $O/pmc_fetch8/run_counter_collection.csv $O/pmc_write8/run_counter_collection.csv --kernel sg_wgrad_fp8_kernel \
  --command "$CMD --conv-dtype fp8 --batch 512 --balance" --out $O/wgrad_fp8_traffic_bs512.json >> $O/traffic.log 2>&1
rm -f $O/pmc_fetch8/run_counter_collection.csv $O/pmc_write8/run_counter_collection.csv
# clock and matrix-pipe occupancy of the conv kernels on one layer (GRBM_GUI_ACTIVE, SQ_VALU_MFMA_BUSY_CYCLES), the bare
# MFMA rates of this device and the constant / per-k-tile split of the DMA-fed kernels
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/clk -o run -- python3 $R/tools/probe_clock.py run > $O/clk.log 2>&1
python3 $R/tools/probe_clock.py report $O/clk/run_counter_collection.csv > $O/probe_clock.txt
rm -rf $O/clk
echo clock probe done
hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak $R/tools/mfma_peak.hip 2>/dev/null && timeout -k 10 240 /tmp/mfma_peak > $O/mfma_peak.txt
echo mfma peak done
timeout -k 10 240 python3 $R/tools/probe_ktile.py > $O/probe_ktile.txt 2>&1
echo profile pass done
