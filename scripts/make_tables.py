"""Markdown rows of the measured table (README, profiles/README.md) from the bench logs of a profiles/rNN directory.
    python scripts/make_tables.py profiles/r04"""
import json, os, sys
d = sys.argv[1]
def line(name):
    p = os.path.join(d, name)
    if not os.path.exists(p):
        return None
    for l in open(p):
        if l.startswith("{"):
            return json.loads(l)
    return None
rows = [("**YelpChi-like, batch 1024 (configs[1]; `bench_yelp.log`)**", "bench_yelp.log"),
        ("the same as the driver runs it (`--steps 20 --warmup 5`; `bench_yelp_driver_style.log`)", "bench_yelp_driver_style.log"),
        ("power-law 2 M nodes / 40 M edges, batch 4096 (`bench_powerlaw_2m.log`)", "bench_powerlaw_2m.log"),
        ("power-law 10 M / 200 M, batch 4096, touched-rows scoring (`bench_powerlaw_10m_200m.log`)", "bench_powerlaw_10m_200m.log"),
        ("the same, whole table scored every step (`PCG_TOUCHED=0`)", "bench_powerlaw_10m_200m_whole_table.log"),
        ("YelpChi-like, emb 128, batch 4096 (configs[2]'s model shape; `bench_yelp_emb128_b4096.log`)", "bench_yelp_emb128_b4096.log"),
        ("Amazon-like, batch 256, rho 0.5 (configs[4]; `bench_amazon.log`)", "bench_amazon.log"),
        ("Amazon-like, rho 0.2", "bench_amazon_rho0.2.log"), ("Amazon-like, rho 0.8", "bench_amazon_rho0.8.log"),
        ("partitioned path, world size 1, RCCL (`--force-partitioned`; `bench_partitioned_w1.log`)", "bench_partitioned_w1.log"),
        ("... one graph per step instead of one per window (`x_partitioned_w1_step_graphs_bench.log`)", "x_partitioned_w1_step_graphs_bench.log"),
        ("... and the all-reduce eager (`x_partitioned_w1_eager_allreduce_bench.log`)", "x_partitioned_w1_eager_allreduce_bench.log"),
        ("partitioned path, world size 1, sharded power-law 10 M / 200 M (`bench_partitioned_w1_powerlaw_10m.log`)", "bench_partitioned_w1_powerlaw_10m.log"),
        ("A/B: the in-kernel sort (`PCG_PRESORT=0`; `x_in_kernel_sort_yelp_bench.log`)", "x_in_kernel_sort_yelp_bench.log"),
        ("A/B: weight-gradient tiles in the select launch (`x_wgrad_in_select_yelp_bench.log`)", "x_wgrad_in_select_yelp_bench.log"),
        ("A/B, timing only: no weight-gradient riders (`x_no_wgrad_riders_timing_only_yelp_bench.log`)", "x_no_wgrad_riders_timing_only_yelp_bench.log")]
print("| workload (bench line, `%s/`) | sampled nodes/s | ms / step | timed call (HIP events), mean [min .. max] | launches | algorithmic | achieved | PMC traffic |" % d)
print("|---|---|---|---|---|---|---|---|")
for label, f in rows:
    j = line(f)
    if j is None:
        continue
    r = j["roofline"]
    ms = r.get("avg_launch_ms")
    mn, mx = r.get("min_launch_ms"), r.get("max_launch_ms")
    rng = f" [{mn * 1e3:.1f} .. {mx * 1e3:.1f}]" if mn and mx else ""
    tr = f"{r['traffic'] / 1e6:.1f} MB" if r.get("traffic") else "—"
    print(f"| {label} | {j['value'] / 1e6:.2f} M | {j['ms_per_step']:.4f} | {ms * 1e3:.1f} µs{rng} | {r.get('launches_timed', '')} | "
          f"{r['algorithmic_bytes_per_launch'] / 1e6:.1f} MB | {r['achieved'] / 1e3:.2f} TB/s ({r['frac'] * 100:.1f} %) | {tr} |")
    c = j.get("cpu_baseline")
    if c and c.get("value"):
        print(f"|   CPU baseline of that line (oracle port, {c.get('cores')} host threads, {str(c.get('sample'))[:60]}) | {c['value']:.0f} | | | | | | |")
    e = j.get("epoch_report")
    if e and f in ("bench_yelp.log", "bench_powerlaw_2m.log", "bench_powerlaw_10m_200m.log"):
        print(f"|   epochs one by one (`epoch_report`, medians of {e.get('epochs')}) | {e['reference_window_nodes_per_s_median'] / 1e6:.2f} M (reference window) / "
              f"{e['pick_inclusive_nodes_per_s_median'] / 1e6:.2f} M (pick-inclusive) | | | | | | |")
