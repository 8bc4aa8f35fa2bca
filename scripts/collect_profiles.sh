#!/bin/bash
# Runs on the GPU box (gpurun): bench lines, rocprofv3 kernel stats and the two PMC passes the numbers in DESIGN.md /
# profiles/README.md come from.  Output under gpurun_out/final/ (copied into profiles/rNN/ afterwards).
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "[collect] bench yelp"; python3 $R/bench.py > $O/bench_yelp.log 2>&1 || exit 1
echo "[collect] kernel trace"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o y -- python3 $R/bench.py --cpu-batches 0 > $O/trace.log 2>&1 || exit 1
echo "[collect] pmc fetch"; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o y -- python3 $R/bench.py --steps 36 --cpu-batches 0 > $O/pmc_fetch.log 2>&1 || exit 1
echo "[collect] pmc write"; rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o y -- python3 $R/bench.py --steps 36 --cpu-batches 0 > $O/pmc_write.log 2>&1 || exit 1
echo "[collect] bench powerlaw 2M"; python3 $R/bench.py --workload powerlaw --nodes 2000000 --edges 40000000 --batch-size 4096 --cpu-batches 1 > $O/bench_powerlaw_2m.log 2>&1 || exit 1
echo "[collect] bench amazon"; python3 $R/bench.py --workload amazon --cpu-batches 4 > $O/bench_amazon.log 2>&1 || exit 1
for rho in 0.2 0.8; do echo "[collect] bench amazon rho $rho"; python3 $R/bench.py --workload amazon --rho $rho --cpu-batches 0 > $O/bench_amazon_rho$rho.log 2>&1 || exit 1; done
echo "[collect] bench emb128 b4096"; python3 $R/bench.py --emb 128 --batch-size 4096 --cpu-batches 1 > $O/bench_yelp_emb128_b4096.log 2>&1 || exit 1
echo "[collect] partitioned path, world size 1"; python3 $R/bench.py --force-partitioned > $O/bench_partitioned_w1.log 2>&1 || exit 1
ls -la $O $O/trace $O/pmc_fetch $O/pmc_write
