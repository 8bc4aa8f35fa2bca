"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes) into
profiles/pmc_traffic.json: HBM-side bytes per select + aggregate call, corrected for gfx950 (FETCH_SIZE counts half
of a 16-B/lane coalesced read: the float4 row gather of gather_chunks is doubled; 4-B/lane kernels are left as read).

    python scripts/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <workload> [out.json] [per_kernel.csv]

PCG_COMMIT (environment): the commit the counters were collected at - stored with the entry, and bench.py prints it next to
`roofline.traffic` (the GPU box has no .git: the collection script is handed the hash).
"""
import csv
import json
import os
import sys
from collections import defaultdict

# the launches of the select + aggregate call.  The training engine's pcg_choose_gather_train = select_rows (+ the label classifier's
# step) and gather_train_kernel (gather + the deferred Adam update + the next step's score pass); no combine_rows (the dense kernel
# finishes multi-chunk sums).  gather_chunks / combine_rows: the calls without the training riders.
CALL_KERNELS = ("select_rows", "select_long_rows", "gather_train_kernel", "gather_chunks", "combine_rows")
# 16 B per lane: FETCH_SIZE x 2.  (gather_train_kernel's Adam workgroups read the gradient slabs 4 B per lane - at most
# n_tiles x n_params x 4 B per launch, 6.9 MB on the YelpChi-like batch: doubled with the rest, so that entry is an upper bound)
WIDE_READERS = ("gather_train_kernel", "gather_chunks")
# PCG_PMC_KERNELS=step: the partitioned path's whole step graph instead (its roofline line is about that one launch): the front
# (score pass: float4 per lane), select, the translating gather (float4), the dense kernel, the weight-gradient GEMMs (float4)
if os.environ.get("PCG_PMC_KERNELS") == "step":
    CALL_KERNELS = ("front_dist_kernel", "select_rows", "select_long_rows", "gather_chunks_dist", "dense_step_kernel", "wgrad_adam_kernel")
    WIDE_READERS = ("front_dist_kernel", "gather_chunks_dist", "wgrad_adam_kernel")


def per_kernel(path, counter):
    tot, n = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            k = row["Kernel_Name"]
            tot[k] += float(row["Counter_Value"])
            n[k] += 1
    return {k: (tot[k] / n[k], n[k]) for k in tot}


def main():
    fetch_csv, write_csv, workload = sys.argv[1:4]
    out = sys.argv[4] if len(sys.argv) > 4 else "profiles/pmc_traffic.json"
    per_csv = sys.argv[5] if len(sys.argv) > 5 else None
    fetch, write = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    if per_csv:
        with open(per_csv, "w") as f:
            f.write("# commit %s\n" % os.environ.get("PCG_COMMIT", "unknown"))
            f.write("kernel,dispatches,FETCH_SIZE_KB_mean,WRITE_SIZE_KB_mean\n")
            for k in sorted(set(fetch) | set(write)):
                f.write('"%s",%d,%.1f,%.1f\n' % (k, fetch.get(k, (0, 0))[1] or write.get(k, (0, 0))[1],
                                                  fetch.get(k, (0, 0))[0], write.get(k, (0, 0))[0]))
    detail, total = {}, 0.0
    for name in CALL_KERNELS:
        fk = sum(v[0] for k, v in fetch.items() if name in k)
        wk = sum(v[0] for k, v in write.items() if name in k)
        mult = 2.0 if name in WIDE_READERS else 1.0
        detail[name] = {"FETCH_SIZE_KB": fk, "fetch_correction": mult, "WRITE_SIZE_KB": wk}
        total += (fk * mult + wk) * 1024.0
    try:
        data = json.load(open(out))
    except Exception:
        data = {}
    data[workload] = {"choose_agg_bytes_per_launch": total, "commit": os.environ.get("PCG_COMMIT", "unknown"), "detail": detail,
                      "correction": "gfx950: FETCH_SIZE reads 1/2 of 16-B/lane coalesced reads (MI355X_MICROARCH.md, HBM) -> "
                                    "gather_train_kernel / gather_chunks (float4 per lane) doubled - an upper bound for the former, whose Adam riders read 4 B per lane; 4-B/lane kernels uncorrected",
                      "note": "mean over the run's dispatches of every kernel of the call; counters in KB"}
    json.dump(data, open(out, "w"), indent=1)
    print(json.dumps(data[workload], indent=1))


if __name__ == "__main__":
    main()
