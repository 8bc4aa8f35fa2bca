#!/bin/bash
# Runs on the GPU box (gpurun): bench lines, rocprofv3 kernel stats and the two PMC passes the numbers in DESIGN.md /
# profiles/README.md come from.  Output under gpurun_out/$TAG/ (copied into profiles/rNN/ afterwards).
#   bash scripts/collect_profiles.sh [TAG] [quick]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PL="--workload powerlaw --nodes 2000000 --edges 40000000 --batch-size 4096"
echo "[collect] bench yelp"; python3 $R/bench.py > $O/bench_yelp.log 2>&1 || exit 1
echo "[collect] kernel trace yelp"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_yelp -o y -- python3 $R/bench.py --cpu-batches 0 > $O/trace_yelp.log 2>&1 || exit 1
echo "[collect] pmc fetch yelp"; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_yelp -o y -- python3 $R/bench.py --steps 36 --cpu-batches 0 > $O/pmc_fetch_yelp.log 2>&1 || exit 1
echo "[collect] pmc write yelp"; rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_yelp -o y -- python3 $R/bench.py --steps 36 --cpu-batches 0 > $O/pmc_write_yelp.log 2>&1 || exit 1
echo "[collect] bench powerlaw 2M"; python3 $R/bench.py $PL --cpu-batches 1 > $O/bench_powerlaw_2m.log 2>&1 || exit 1
echo "[collect] kernel trace powerlaw 2M"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_pl2m -o y -- python3 $R/bench.py $PL --steps 60 --cpu-batches 0 > $O/trace_pl2m.log 2>&1 || exit 1
echo "[collect] pmc fetch powerlaw 2M"; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_pl2m -o y -- python3 $R/bench.py $PL --steps 24 --cpu-batches 0 > $O/pmc_fetch_pl2m.log 2>&1 || exit 1
echo "[collect] pmc write powerlaw 2M"; rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_pl2m -o y -- python3 $R/bench.py $PL --steps 24 --cpu-batches 0 > $O/pmc_write_pl2m.log 2>&1 || exit 1
echo "[collect] bench amazon"; python3 $R/bench.py --workload amazon --cpu-batches 4 > $O/bench_amazon.log 2>&1 || exit 1
for rho in 0.2 0.8; do echo "[collect] bench amazon rho $rho"; python3 $R/bench.py --workload amazon --rho $rho --cpu-batches 0 > $O/bench_amazon_rho$rho.log 2>&1 || exit 1; done
echo "[collect] bench emb128 b4096"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_e128 -o y -- python3 $R/bench.py --emb 128 --batch-size 4096 --cpu-batches 0 > $O/bench_yelp_emb128_b4096.log 2>&1 || exit 1
echo "[collect] partitioned path, world size 1"; python3 $R/bench.py --force-partitioned --cpu-batches 0 > $O/bench_partitioned_w1.log 2>&1 || exit 1
if [ "$2" != "quick" ]; then
echo "[collect] partitioned path, world size 1, sharded power-law 10M / 200M"; python3 $R/bench.py --force-partitioned --workload powerlaw --nodes 10000000 --edges 200000000 --batch-size 4096 --steps 40 --cpu-batches 0 > $O/bench_partitioned_w1_powerlaw_10m.log 2>&1 || exit 1
echo "[collect] bench powerlaw 10M / 200M"; python3 $R/bench.py --workload powerlaw --nodes 10000000 --edges 200000000 --batch-size 4096 --steps 60 --cpu-batches 0 > $O/bench_powerlaw_10m_200m.log 2>&1 || exit 1
echo "[collect] kernel trace powerlaw 10M / 200M"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_pl10m -o y -- python3 $R/bench.py --workload powerlaw --nodes 10000000 --edges 200000000 --batch-size 4096 --steps 40 --cpu-batches 0 > $O/trace_pl10m.log 2>&1 || exit 1
fi
ls $O
