#!/bin/bash
# Round evidence in one gpurun call: the bench line, rocprofv3 --kernel-trace --stats of the same command, the PMC
# passes for K1's HBM traffic, and the 3-D slab's kernel stats.  Outputs under gpurun_out/prof_round/ (copy the
# summaries into profiles/).  usage: tools/prof_round.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_round
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $ROOT
echo "[0] PMC passes (first: bench.py quotes the traffic recorded for THESE sources)"; BENCH_ARGS="--steps 4 --warmup 2" tools/prof_pmc.sh gpurun_out/prof_round/pmc
python3 tools/pmc_summary.py gpurun_out/prof_round/pmc k_push_deposit_tiled_2d > $OUT/pmc_k1.txt
python3 tools/make_traffic_json.py gpurun_out/prof_round/pmc > $OUT/traffic.log 2>&1
cp profiles/r04_k1_traffic.json $OUT/ 2>/dev/null
echo "[0b] PMC passes (3-D slab)"; tools/prof_pmc3d.sh gpurun_out/prof_round/pmc3d
python3 tools/pmc_summary.py gpurun_out/prof_round/pmc3d k_push_deposit_tiled_3d > $OUT/pmc_k13d.txt
python3 tools/make_traffic_json.py gpurun_out/prof_round/pmc3d 3d > $OUT/traffic3d.log 2>&1
cp profiles/r04_k13d_traffic.json $OUT/ 2>/dev/null
echo "[0c] PMC passes (C5 slab leg, e- + p in one launch)"; tools/prof_pmc_c5.sh gpurun_out/prof_round/pmc_c5
python3 tools/pmc_summary.py gpurun_out/prof_round/pmc_c5 k_push_deposit_tiled_3d > $OUT/pmc_k13d_c5.txt
python3 tools/make_traffic_json.py gpurun_out/prof_round/pmc_c5 c5 > $OUT/traffic_c5.log 2>&1
cp profiles/r04_k13d_c5_traffic.json $OUT/ 2>/dev/null
cd /tmp
echo "[1] bench.py"; timeout -k 10 500 python3 $ROOT/bench.py > $OUT/bench.json.log 2> $OUT/bench.err; echo "   exit $?"
echo "[1b] bench.py --steps 100 --warmup 10 (SURVEY 8d step counts)"; timeout -k 10 300 python3 $ROOT/bench.py --steps 100 --warmup 10 --no-extra --no-cpu-baseline > $OUT/bench_100steps.json.log 2>> $OUT/bench.err; echo "   exit $?"
echo "[2] kernel stats (2-D)"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats2d -- python3 $ROOT/bench.py --no-cpu-baseline --no-extra > $OUT/stats2d.log 2>&1; grep '^{"metric' $OUT/stats2d.log | tail -1 > $OUT/bench_under_rocprof.json.log; echo "   exit $?"
echo "[3] kernel stats (3-D slab)"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats3d -- python3 $ROOT/tools/bench3d.py > $OUT/stats3d.log 2>&1; echo "   exit $?"
echo "[4] kernel stats (the C5-slab leg of bench.py: e- + p in one K1-3D launch)"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5leg -- python3 $ROOT/tools/bench_c5leg.py 40 12 > $OUT/stats_c5leg.log 2>&1; echo "   exit $?"
echo "[5] 3-D orders (striped / padded), LDS counters"; for o in striped padded; do timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc3d_$o -- python3 $ROOT/tools/bench3d.py --order $o --steps 10 --warmup 12 > $OUT/pmc3d_$o.log 2>&1; echo "== $o" >> $OUT/pmc3d_orders.txt; grep '^{' $OUT/pmc3d_$o.log | tail -1 | cut -c1-200 >> $OUT/pmc3d_orders.txt; python3 $ROOT/tools/pmc_summary.py $OUT/pmc3d_$o k_push_deposit_tiled_3d >> $OUT/pmc3d_orders.txt; done
find $OUT -name "*kernel_stats.csv" | head
