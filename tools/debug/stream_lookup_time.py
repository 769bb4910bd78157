import os, sys, argparse, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
args = argparse.Namespace(bases=1_387_536_274, k=31, m=21, mean_len=85.0, canonical=False, seed=0x5555AAAA, cache_dir="/tmp", verbose=False)
d, path = bench.get_index(args, 0, 1, lambda: None)
d.to_device(0)
dev = torch.device("cuda", 0)
R, L = 2_000_000, 150
rng = np.random.default_rng(1)
reads = np.frombuffer(b"ACTG", dtype=np.uint8)[rng.integers(0, 4, (R, L), dtype=np.uint8)]
d_bases = torch.from_numpy(reads.reshape(-1)).to(dev)
d_off = torch.from_numpy((np.arange(R + 1, dtype=np.uint64) * np.uint64(L)).view(np.int64)).to(dev)
ids = torch.empty(R * L, dtype=torch.int64, device=dev)
rep = torch.zeros(6, dtype=torch.int64, device=dev)
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d.streaming_lookup_device(0, d_bases.data_ptr(), d_off.data_ptr(), R, R * L, ids.data_ptr(), d_report=rep.data_ptr())
    torch.cuda.synchronize(); print("call", i, round((time.perf_counter() - t0) * 1e3, 2), "ms", flush=True)
