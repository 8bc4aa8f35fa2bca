import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/*.log")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-40s %8.3f M/s  %7.2f us/step  call %6.2f us  frac %.3f" % (f.split("/")[-1], d["value"] / 1e6, d["ms_per_step"] * 1e3, d["roofline"]["avg_launch_ms"] * 1e3, d["roofline"]["frac"]))
    except Exception as e:
        print(f, "??", e)
