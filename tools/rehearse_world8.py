"""World-8 rehearsal of the sharding + gather helpers on CPU (gloo): N = 10 000 (C4), a ragged N and N < world."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist, torch.multiprocessing as mp
from openglottal_amd.dist import shard_range, all_gather_areas, all_gather_rows
def work(rank, world, n):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n, rank, world)
    full = np.arange(n, dtype=np.int32) * 3 + 1
    got = all_gather_areas(torch.from_numpy(full[lo:hi].copy()), n)
    rows = all_gather_rows(torch.from_numpy(np.stack([full[lo:hi]] * 5, 1).astype(np.float32)), n, 5)
    assert np.array_equal(got.numpy(), full), (rank, got[:10])
    assert np.array_equal(rows.numpy()[:, 2], full.astype(np.float32))
    dist.destroy_process_group()
if __name__ == "__main__":
    for n in (10000, 10003, 7):
        mp.spawn(work, args=(8, n), nprocs=8, join=True)
        print("world 8, N =", n, "ok")
