#!/bin/bash
# Collect the round's profile summaries on the GPU box: rocprofv3 kernel trace + stats of the bench command, the PMC
# traffic passes (FETCH_SIZE, WRITE_SIZE: separate runs, --kernel-trace only) and the SQ instruction-mix passes.
# Raw CSVs are summarised here and deleted (they exceed what gpurun copies back).  usage: tools/profile_round.sh rNN
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/profile_$TAG
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $OUT/stats_bench_line.json 2> $OUT/stats.err
T=$(ls $OUT/stats/*/*kernel_trace.csv | head -1); S=$(ls $OUT/stats/*/*kernel_stats.csv | head -1)
python3 $R/tools/profile_summary.py trace $T > $OUT/${TAG}_bench_by_grid.body.md
cp $S $OUT/${TAG}_bench_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/write.json 2> $OUT/write.err
python3 $R/tools/profile_summary.py pmc $(ls $OUT/fetch/*/*counter_collection.csv | head -1) $(ls $OUT/write/*/*counter_collection.csv | head -1) > $OUT/${TAG}_pmc_traffic.json
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d $OUT/mixa --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/mixa.json 2> $OUT/mixa.err
rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES SQ_WAIT_ANY SQ_WAVE_CYCLES -d $OUT/mixb --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/mixb.json 2> $OUT/mixb.err
python3 $R/tools/pmc_instruction_mix.py $(ls $OUT/mixa/*/*counter_collection.csv | head -1) $(ls $OUT/mixb/*/*counter_collection.csv | head -1) icp2_fused icp2_resume icp2_wide icp2_far prep_targets voxel_small nn_batch rotation_scores rotation_search_batch ray_ > $OUT/${TAG}_pmc_instruction_mix.json
rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/mixa $OUT/mixb
# the bench line reads the traffic summary from profiles/ (and refuses one measured on other kernel sources)
cp $OUT/${TAG}_pmc_traffic.json $R/profiles/${TAG}_pmc_traffic.json
cp $OUT/${TAG}_pmc_instruction_mix.json $R/profiles/${TAG}_pmc_instruction_mix.json
cd $R && python3 bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/bench.err
ls -la $OUT
