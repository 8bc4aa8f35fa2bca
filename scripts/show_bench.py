"""one line per bench log: python scripts/show_bench.py gpurun_out/x/*.log"""
import json, sys
for f in sys.argv[1:]:
    for line in open(f):
        if line.startswith('{"metric"'):
            d = json.loads(line); r = d["roofline"]
            print("%-60s %7.3f M/s %7.4f ms/step | call avg %6.2f us min %6.2f max %6.2f n=%d frac %.3f alg %.1f MB" % (
                f.split("/")[-1], d["value"] / 1e6, d["ms_per_step"], r["avg_launch_ms"] * 1e3, r.get("min_launch_ms", 0) * 1e3,
                r.get("max_launch_ms", 0) * 1e3, r["launches_timed"], r["frac"], r["algorithmic_bytes_per_launch"] / 1e6))
