O=gpurun_out/$1; shift
mkdir -p $O
C="--cpu-batches 0 --report-epochs 0 --verify-batches 0"
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gpu.log 2>&1 || exit 1
timeout -k 10 200 python scripts/stamp_probe.py > $O/probe.log 2>&1 || exit 1
timeout -k 10 240 python3 bench.py $C > $O/yelp.log 2>&1 || exit 1
timeout -k 10 240 python3 bench.py $C --emb 128 --batch-size 4096 > $O/e128.log 2>&1 || exit 1
timeout -k 10 240 python3 bench.py $C --steps 20 --warmup 5 > $O/yelp_driver.log 2>&1 || exit 1
