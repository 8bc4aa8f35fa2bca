O=gpurun_out/$1; shift
mkdir -p $O
C="--cpu-batches 0 --report-epochs 0 --verify-batches 0"
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "touched" > $O/gpu.log 2>&1 || exit 1
timeout -k 10 500 python3 bench.py $C --workload powerlaw --nodes 10000000 --edges 200000000 --batch-size 4096 --steps 60 > $O/pl10m.log 2>&1 || exit 1
