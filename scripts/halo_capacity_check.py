"""CPU check of the partitioned path's default halo capacities: for world sizes 2 / 4 / 8 and every rank, the distinct remote
neighbours (per owner) of a window of `window` x `batch` picked centres vs the pitch DistributedPCGNN would reserve."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pcgnn_amd import synth
from pcgnn_amd.dist import HaloExchange, Partition, expected_halo_rows, shard_pick_weights, shard_workload, total_degree

def check(w, B, window, worlds=(2, 4, 8), trials=3):
    for world in worlds:
        worst = 0.0
        pitches = []
        info = []
        for rank in range(world):
            part = Partition.balanced(total_degree(w.csr), world, rank)
            sh = shard_workload(w, part)
            y_all = w.labels[w.idx_train]
            wts = shard_pick_weights(w.labels[sh["idx_train_local"]], sh["homo_deg_train"], len(y_all), int(y_all.sum()))
            deg_rel = [np.diff(ip)[sh["idx_train_local"] - part.lo] for ip, _ in sh["csr"]]
            halo_rows = expected_halo_rows(deg_rel, wts, B * window, world, w.n - part.n_local, kept=1.0)
            pitch = min(part.n_max, -(-halo_rows * 5 // (4 * max(world - 1, 1))) + 64)
            pitches.append(pitch)
            info.append((part, sh, wts))
        pitch = max(pitches)
        if any(len(sh["idx_train_local"]) == 0 for _, sh, _ in info):
            print(f"{w.name}: world {world}: a rank owns no training node (DistributedPCGNN refuses this partition)")
            continue
        for rank, (part, sh, wts) in enumerate(info):
            P = len(sh["train_pos"])
            X_ext = torch.zeros(part.n_local + P + (world - 1) * pitch, 4)
            hx = HaloExchange(part, X_ext, sh["train_pos"], pitch)
            csr_t = [(torch.from_numpy(ip), torch.from_numpy(ix.astype(np.int64))) for ip, ix in sh["csr"]]
            rs = np.random.RandomState(rank)
            p = wts / wts.sum()
            for t in range(trials):
                centres = rs.choice(sh["idx_train_local"] - part.lo, size=B * window, p=p)
                hx.collect(csr_t, torch.from_numpy(centres.astype(np.int32)))
                seen = hx.max_seen
                worst = max(worst, seen["rows_from_one_owner"] / pitch)
                assert int(hx.overflow_word) == 0, (world, rank, seen, pitch)
        print(f"{w.name}: world {world} window {window} x batch {B}: pitch {pitch} (n_max {info[0][0].n_max}), worst fill {worst:.2f}")

if __name__ == "__main__":
    check(synth.yelp_like(0), 1024, 8)
    check(synth.yelp_like(0), 4096, 8)
    check(synth.amazon_like(0), 256, 8)
    check(synth.power_law(200000, 4000000, 0), 4096, 8)
