cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2_ab; mkdir -p $O
for rep in 1 2; do for nb in 768 512 256; do
PCG_SEL_BLOCKS=$nb rocprofv3 --kernel-trace --stats --output-format csv -d $O/y_${nb}_$rep -o y -- python3 $R/bench.py --cpu-batches 0 --steps 120 > $O/y_${nb}_$rep.log 2>&1
done; done
echo done
