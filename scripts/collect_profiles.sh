#!/bin/bash
# Runs on the GPU box (gpurun): bench lines, rocprofv3 kernel stats and the two PMC passes the numbers in DESIGN.md /
# profiles/README.md come from.  Output under gpurun_out/$TAG/ (copied into profiles/rNN/ afterwards).
#   PCG_COMMIT=<hash of the commit being profiled> bash scripts/collect_profiles.sh [TAG] [quick|part|big]
#   quick: single-GPU YelpChi-like / Amazon-like / emb 128 / power-law 2 M;  part: the partitioned path + the A/B runs;
#   big: the 10 M-node graph (three calls fit gpurun's 20-minute limit each)
# For every workload the PMC passes come FIRST and their summary is copied to profiles/pmc_traffic.json, so that the bench lines of
# the same call quote this commit's counters.  (The box has no .git: every stats file gets a `# commit` line from PCG_COMMIT.)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
MODE=${2:-quick}
O=$R/gpurun_out/$TAG
mkdir -p $O
export PCG_COMMIT=${PCG_COMMIT:-unknown}
echo "$PCG_COMMIT" > $O/COMMIT
cd /tmp && export TMPDIR=/tmp
stamp() { for f in "$@"; do [ -f "$f" ] && sed -i "1i # commit $PCG_COMMIT" "$f"; done; }
# the traffic summary accumulates over the calls: start from what the repository holds
[ -f $O/pmc_traffic.json ] || cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json 2>/dev/null || true
pmc() {  # pmc <name> <json key> <per-kernel csv suffix> <bench args...>
  local name=$1 key=$2 suf=$3; shift 3
  echo "[collect] pmc fetch $name"; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$name -o y -- python3 $R/bench.py "$@" > $O/pmc_fetch_$name.log 2>&1 || return 1
  echo "[collect] pmc write $name"; rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$name -o y -- python3 $R/bench.py "$@" > $O/pmc_write_$name.log 2>&1 || return 1
  python3 $R/scripts/pmc_traffic.py $O/pmc_fetch_$name/y_counter_collection.csv $O/pmc_write_$name/y_counter_collection.csv $key $O/pmc_traffic.json $O/pmc_fetch_write_per_kernel_$suf.csv > $O/pmc_traffic_$name.log 2>&1 || return 1
  rm -rf $O/pmc_fetch_$name $O/pmc_write_$name $O/pmc_fetch_$name.log $O/pmc_write_$name.log
  cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json
}
trace() {  # trace <name> <bench args...>
  local name=$1; shift
  echo "[collect] kernel trace $name"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$name -o y -- python3 $R/bench.py "$@" > $O/trace_$name.log 2>&1 || return 1
  cp $O/trace_$name/y_kernel_stats.csv $O/kernel_stats_$name.csv && stamp $O/kernel_stats_$name.csv
  rm -rf $O/trace_$name
}
NOX="--cpu-batches 0 --report-epochs 0 --verify-batches 0 --post-brackets 0"
PL="--workload powerlaw --nodes 2000000 --edges 40000000 --batch-size 4096"
PL10="--workload powerlaw --nodes 10000000 --edges 200000000 --batch-size 4096"
if [ "$MODE" = "quick" ]; then
echo "[collect] smoke"; (cd $R && python3 -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1 || exit 1
pmc yelp yelp yelp --steps 36 $NOX || exit 1
echo "[collect] bench yelp"; python3 $R/bench.py > $O/bench_yelp.log 2>&1 || exit 1
echo "[collect] bench yelp, as the driver runs it"; python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_yelp_driver_style.log 2>&1 || exit 1
trace yelp --cpu-batches 0 || exit 1
pmc pl2m powerlaw_2000000_40000000_b4096 powerlaw_2m $PL --steps 24 $NOX || exit 1
echo "[collect] bench powerlaw 2M"; python3 $R/bench.py $PL --cpu-batches 1 > $O/bench_powerlaw_2m.log 2>&1 || exit 1
trace pl2m $PL --steps 60 --cpu-batches 0 --report-epochs 0 --verify-batches 0 || exit 1
pmc amazon amazon amazon --workload amazon --steps 36 $NOX || exit 1
echo "[collect] bench amazon"; python3 $R/bench.py --workload amazon --cpu-batches 4 > $O/bench_amazon.log 2>&1 || exit 1
for rho in 0.2 0.8; do echo "[collect] bench amazon rho $rho"; python3 $R/bench.py --workload amazon --rho $rho --cpu-batches 0 > $O/bench_amazon_rho$rho.log 2>&1 || exit 1; done
pmc e128 yelp_b4096 e128 --emb 128 --batch-size 4096 --steps 24 $NOX || exit 1
echo "[collect] bench emb128 b4096"; python3 $R/bench.py --emb 128 --batch-size 4096 --cpu-batches 0 > $O/bench_yelp_emb128_b4096.log 2>&1 || exit 1
trace e128 --emb 128 --batch-size 4096 --cpu-batches 0 --report-epochs 0 --verify-batches 0 || exit 1
fi
if [ "$MODE" = "part" ]; then
PW="--force-partitioned --steps 48 --cpu-batches 0"
export PCG_DIST_WINDOW_GRAPH=0; PCG_PMC_KERNELS=step pmc part yelp_partitioned_w1 partitioned $PW || exit 1; unset PCG_DIST_WINDOW_GRAPH
echo "[collect] partitioned path, world size 1"; python3 $R/bench.py --force-partitioned --cpu-batches 0 > $O/bench_partitioned_w1.log 2>&1 || exit 1
echo "[collect] partitioned path: one graph per step, and with the all-reduce eager as well (round 3's host calls)"
PCG_DIST_WINDOW_GRAPH=0 python3 $R/bench.py --force-partitioned --cpu-batches 0 > $O/x_partitioned_w1_step_graphs_bench.log 2>&1 || exit 1
PCG_DIST_GRAPH_COLLECTIVES=0 python3 $R/bench.py --force-partitioned --cpu-batches 0 > $O/x_partitioned_w1_eager_allreduce_bench.log 2>&1 || exit 1
trace part --force-partitioned --cpu-batches 0 || exit 1
echo "[collect] two ranks on one GPU (gloo-staged collectives: plumbing only)"; PCG_BENCH_BACKEND=gloo python3 $R/bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_gloo_2ranks.log 2>&1 || exit 1
echo "[collect] A/B: in-kernel sort (PCG_PRESORT=0), weight-gradient tiles in the select launch (PCG_WGRAD_IN_SELECT=1), no weight-gradient riders at all (timing only)"
PCG_PRESORT=0 python3 $R/bench.py $NOX > $O/x_in_kernel_sort_yelp_bench.log 2>&1 || exit 1
PCG_PRESORT=0 trace yelp_in_kernel_sort $NOX || exit 1; mv $O/kernel_stats_yelp_in_kernel_sort.csv $O/x_in_kernel_sort_yelp_kernel_stats.csv; rm -f $O/trace_yelp_in_kernel_sort.log
PCG_WGRAD_IN_SELECT=1 python3 $R/bench.py $NOX > $O/x_wgrad_in_select_yelp_bench.log 2>&1 || exit 1
PCG_WGRAD_OFF=1 python3 $R/bench.py $NOX > $O/x_no_wgrad_riders_timing_only_yelp_bench.log 2>&1 || exit 1
echo "[collect] sort probe"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_sort -o y -- python3 $R/scripts/sort_probe.py > $O/sort_probe.log 2>&1 || exit 1
cp $O/trace_sort/y_kernel_stats.csv $O/kernel_stats_sort_probe.csv && stamp $O/kernel_stats_sort_probe.csv; rm -rf $O/trace_sort
echo "[collect] partitioned path, world size 1, sharded power-law 10M / 200M"; python3 $R/bench.py --force-partitioned $PL10 --steps 40 --cpu-batches 0 > $O/bench_partitioned_w1_powerlaw_10m.log 2>&1 || exit 1
fi
if [ "$MODE" = "amazon" ]; then       # (the Amazon-like lines on their own: PMC passes, then the three bench lines)
pmc amazon amazon amazon --workload amazon --steps 36 $NOX || exit 1
echo "[collect] bench amazon"; python3 $R/bench.py --workload amazon --cpu-batches 4 > $O/bench_amazon.log 2>&1 || exit 1
for rho in 0.2 0.8; do echo "[collect] bench amazon rho $rho"; python3 $R/bench.py --workload amazon --rho $rho --cpu-batches 0 > $O/bench_amazon_rho$rho.log 2>&1 || exit 1; done
fi
if [ "$MODE" = "big" ]; then
pmc pl10m powerlaw_10000000_200000000_b4096 powerlaw_10m $PL10 --steps 20 $NOX || exit 1
echo "[collect] bench powerlaw 10M / 200M"; python3 $R/bench.py $PL10 --steps 60 --cpu-batches 0 > $O/bench_powerlaw_10m_200m.log 2>&1 || exit 1
trace pl10m $PL10 --steps 40 --cpu-batches 0 --report-epochs 0 --verify-batches 0 || exit 1
echo "[collect] bench powerlaw 10M / 200M, whole table scored"; PCG_TOUCHED=0 python3 $R/bench.py $PL10 --steps 60 $NOX > $O/bench_powerlaw_10m_200m_whole_table.log 2>&1 || exit 1
fi
ls $O
