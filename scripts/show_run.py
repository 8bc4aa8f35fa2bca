"""Print the headline numbers of a gpurun_out/<tag>/ collection: bench lines + per-kernel averages of the trace."""
import csv, json, sys, os
d = sys.argv[1]
for f in sorted(os.listdir(d)):
    if f.startswith("bench") and f.endswith(".log"):
        try:
            j = json.loads(open(os.path.join(d, f)).read().strip().splitlines()[-1])
        except Exception as e:
            print(f, "unreadable", e); continue
        r = j.get("roofline", {})
        print(f"{f:32s} {j['value']/1e6:7.2f} M nodes/s  {j['ms_per_step']*1e3:7.2f} us/step  sel+gather {r.get('avg_launch_ms', 0)*1e3:6.2f} us frac {r.get('frac', 0):.3f}")
        if "epoch_report" in j:
            e = j["epoch_report"]
            print(f"{'':32s} epoch window {e['reference_window_ms_per_epoch_median']*1e3:.1f} us, with pick {e['pick_inclusive_ms_per_epoch_median']*1e3:.1f} us")
for sub in sorted(os.listdir(d)):
    p = os.path.join(d, sub)
    if os.path.isdir(p):
        for f in os.listdir(p):
            if f.endswith("kernel_stats.csv"):
                print("--", sub)
                for r in list(csv.DictReader(open(os.path.join(p, f))))[:12]:
                    print(f"   {r['Name'][:60]:60s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us  {r['Percentage']:>6s} %")
