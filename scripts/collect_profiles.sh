#!/bin/bash
# Runs on the GPU box (gpurun): bench lines, rocprofv3 kernel stats and the two PMC passes the numbers in DESIGN.md /
# profiles/README.md come from.  Output under gpurun_out/$TAG/ (copied into profiles/rNN/ afterwards).
#   PCG_COMMIT=<hash of the commit being profiled> bash scripts/collect_profiles.sh [TAG] [quick|big]
#   quick: everything but the 10 M-node graph; big: the 10 M-node graph only (two calls fit gpurun's 20-minute limit)
# (the box has no .git: every stats file gets a `# commit` line / a COMMIT file from PCG_COMMIT)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
O=$R/gpurun_out/$TAG
mkdir -p $O
export PCG_COMMIT=${PCG_COMMIT:-unknown}
echo "$PCG_COMMIT" > $O/COMMIT
cd /tmp && export TMPDIR=/tmp
stamp() { for f in "$@"; do [ -f "$f" ] && sed -i "1i # commit $PCG_COMMIT" "$f"; done; }
PL="--workload powerlaw --nodes 2000000 --edges 40000000 --batch-size 4096"
PL10="--workload powerlaw --nodes 10000000 --edges 200000000 --batch-size 4096"
if [ "$2" != "big" ]; then
echo "[collect] smoke"; (cd $R && python3 -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1 || exit 1
echo "[collect] bench yelp"; python3 $R/bench.py > $O/bench_yelp.log 2>&1 || exit 1
echo "[collect] bench yelp, as the driver runs it"; python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_yelp_driver_style.log 2>&1 || exit 1
echo "[collect] kernel trace yelp"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_yelp -o y -- python3 $R/bench.py --cpu-batches 0 > $O/trace_yelp.log 2>&1 || exit 1
stamp $O/trace_yelp/y_kernel_stats.csv
echo "[collect] pmc fetch yelp"; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_yelp -o y -- python3 $R/bench.py --steps 36 --cpu-batches 0 --report-epochs 0 --verify-batches 0 > $O/pmc_fetch_yelp.log 2>&1 || exit 1
echo "[collect] pmc write yelp"; rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_yelp -o y -- python3 $R/bench.py --steps 36 --cpu-batches 0 --report-epochs 0 --verify-batches 0 > $O/pmc_write_yelp.log 2>&1 || exit 1
python3 $R/scripts/pmc_traffic.py $O/pmc_fetch_yelp/y_counter_collection.csv $O/pmc_write_yelp/y_counter_collection.csv yelp $O/pmc_traffic.json $O/pmc_fetch_write_per_kernel_yelp.csv > $O/pmc_traffic_yelp.log 2>&1 || exit 1
rm -rf $O/pmc_fetch_yelp $O/pmc_write_yelp
echo "[collect] bench powerlaw 2M"; python3 $R/bench.py $PL --cpu-batches 1 > $O/bench_powerlaw_2m.log 2>&1 || exit 1
echo "[collect] kernel trace powerlaw 2M"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_pl2m -o y -- python3 $R/bench.py $PL --steps 60 --cpu-batches 0 --report-epochs 0 --verify-batches 0 > $O/trace_pl2m.log 2>&1 || exit 1
stamp $O/trace_pl2m/y_kernel_stats.csv
echo "[collect] pmc fetch powerlaw 2M"; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_pl2m -o y -- python3 $R/bench.py $PL --steps 24 --cpu-batches 0 --report-epochs 0 --verify-batches 0 > $O/pmc_fetch_pl2m.log 2>&1 || exit 1
echo "[collect] pmc write powerlaw 2M"; rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_pl2m -o y -- python3 $R/bench.py $PL --steps 24 --cpu-batches 0 --report-epochs 0 --verify-batches 0 > $O/pmc_write_pl2m.log 2>&1 || exit 1
python3 $R/scripts/pmc_traffic.py $O/pmc_fetch_pl2m/y_counter_collection.csv $O/pmc_write_pl2m/y_counter_collection.csv powerlaw_2000000_40000000_b4096 $O/pmc_traffic.json $O/pmc_fetch_write_per_kernel_powerlaw_2m.csv > $O/pmc_traffic_pl2m.log 2>&1 || exit 1
rm -rf $O/pmc_fetch_pl2m $O/pmc_write_pl2m
echo "[collect] bench amazon"; python3 $R/bench.py --workload amazon --cpu-batches 4 > $O/bench_amazon.log 2>&1 || exit 1
for rho in 0.2 0.8; do echo "[collect] bench amazon rho $rho"; python3 $R/bench.py --workload amazon --rho $rho --cpu-batches 0 > $O/bench_amazon_rho$rho.log 2>&1 || exit 1; done
echo "[collect] bench emb128 b4096"; python3 $R/bench.py --emb 128 --batch-size 4096 --cpu-batches 0 > $O/bench_yelp_emb128_b4096.log 2>&1 || exit 1
echo "[collect] kernel trace emb128 b4096"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_e128 -o y -- python3 $R/bench.py --emb 128 --batch-size 4096 --cpu-batches 0 --report-epochs 0 --verify-batches 0 > $O/trace_e128.log 2>&1 || exit 1
stamp $O/trace_e128/y_kernel_stats.csv
E128="--emb 128 --batch-size 4096 --steps 24 --cpu-batches 0 --report-epochs 0 --verify-batches 0 --post-brackets 0"
echo "[collect] pmc fetch emb128 b4096"; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_e128 -o y -- python3 $R/bench.py $E128 > $O/pmc_fetch_e128.log 2>&1 || exit 1
echo "[collect] pmc write emb128 b4096"; rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_e128 -o y -- python3 $R/bench.py $E128 > $O/pmc_write_e128.log 2>&1 || exit 1
python3 $R/scripts/pmc_traffic.py $O/pmc_fetch_e128/y_counter_collection.csv $O/pmc_write_e128/y_counter_collection.csv yelp_b4096 $O/pmc_traffic.json $O/pmc_fetch_write_per_kernel_e128.csv > $O/pmc_traffic_e128.log 2>&1 || exit 1
rm -rf $O/pmc_fetch_e128 $O/pmc_write_e128
echo "[collect] stream / pipelining probes (negative results: scripts/pipe_probe.py)"; python3 $R/scripts/pipe_probe.py > $O/x_decoupled_phases_yelp_probe.log 2>&1 || exit 1
PROBE_STREAMS=1 python3 $R/scripts/pipe_probe.py > $O/x_stream_pipeline_yelp_probe.log 2>&1 || exit 1
echo "[collect] partitioned path, world size 1"; python3 $R/bench.py --force-partitioned --cpu-batches 0 > $O/bench_partitioned_w1.log 2>&1 || exit 1
echo "[collect] kernel trace partitioned path"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_part -o y -- python3 $R/bench.py --force-partitioned --cpu-batches 0 > $O/trace_part.log 2>&1 || exit 1
stamp $O/trace_part/y_kernel_stats.csv
echo "[collect] two ranks on one GPU (gloo-staged collectives: plumbing only)"; PCG_BENCH_BACKEND=gloo python3 $R/bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_gloo_2ranks.log 2>&1 || exit 1
fi
if [ "$2" != "quick" ]; then
echo "[collect] bench powerlaw 10M / 200M"; python3 $R/bench.py $PL10 --steps 60 --cpu-batches 0 > $O/bench_powerlaw_10m_200m.log 2>&1 || exit 1
echo "[collect] kernel trace powerlaw 10M / 200M"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_pl10m -o y -- python3 $R/bench.py $PL10 --steps 40 --cpu-batches 0 --report-epochs 0 --verify-batches 0 > $O/trace_pl10m.log 2>&1 || exit 1
stamp $O/trace_pl10m/y_kernel_stats.csv
P10="$PL10 --steps 20 --cpu-batches 0 --report-epochs 0 --verify-batches 0 --post-brackets 0"
echo "[collect] pmc fetch powerlaw 10M"; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_pl10m -o y -- python3 $R/bench.py $P10 > $O/pmc_fetch_pl10m.log 2>&1 || exit 1
echo "[collect] pmc write powerlaw 10M"; rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_pl10m -o y -- python3 $R/bench.py $P10 > $O/pmc_write_pl10m.log 2>&1 || exit 1
python3 $R/scripts/pmc_traffic.py $O/pmc_fetch_pl10m/y_counter_collection.csv $O/pmc_write_pl10m/y_counter_collection.csv powerlaw_10000000_200000000_b4096 $O/pmc_traffic.json $O/pmc_fetch_write_per_kernel_powerlaw_10m.csv > $O/pmc_traffic_pl10m.log 2>&1 || exit 1
rm -rf $O/pmc_fetch_pl10m $O/pmc_write_pl10m
echo "[collect] bench powerlaw 10M / 200M, whole table scored"; PCG_TOUCHED=0 python3 $R/bench.py $PL10 --steps 60 --cpu-batches 0 --report-epochs 0 --verify-batches 0 > $O/bench_powerlaw_10m_200m_whole_table.log 2>&1 || exit 1
echo "[collect] partitioned path, world size 1, sharded power-law 10M / 200M"; python3 $R/bench.py --force-partitioned $PL10 --steps 40 --cpu-batches 0 > $O/bench_partitioned_w1_powerlaw_10m.log 2>&1 || exit 1
fi
# keep the summaries, drop the raw traces (tens of MB)
for d in trace_yelp trace_pl2m trace_e128 trace_part trace_pl10m; do
  [ -f $O/$d/y_kernel_stats.csv ] && cp $O/$d/y_kernel_stats.csv $O/kernel_stats_${d#trace_}.csv
  rm -rf $O/$d
done
ls $O
