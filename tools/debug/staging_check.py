import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sshash_amd
from sshash_amd import _binding
from oracle.ground_truth import GroundTruth, read_fasta_sequences
fasta = os.path.join(ROOT, "tests/golden/salmonella_enterica_k31_ust.fa.gz")
k, m = 31, 13
d = sshash_amd.Dictionary.build(fasta, k=k, m=m, num_threads=0); d.to_device(0)
gt = GroundTruth(read_fasta_sequences(fasta, k), k); n = gt.num_kmers
ids = np.arange(n, dtype=np.uint64); q = gt.kmers(ids)
lib = _binding._load()
out8 = (C.c_ulonglong * 8)()
dq = torch.from_numpy(q.view(np.int64)).cuda(); out = torch.empty(n, dtype=torch.int64, device="cuda")
mo = torch.zeros(n, dtype=torch.uint8, device="cuda")
for rep in range(3):
    lib.sshash_debug_counters(out8, 1)
    d.lookup_device(0, dq.data_ptr(), n, out.data_ptr()); torch.cuda.synchronize()
    lib.sshash_debug_counters(out8, 1)
    bad = int((out.cpu().numpy().view(np.uint64) != ids).sum())
    print("ids   rep", rep, "bad", bad, "checked", out8[0], "LDS != global", out8[1], "of which final miss", out8[2])
    d.is_member_device(0, dq.data_ptr(), n, mo.data_ptr()); torch.cuda.synchronize()
    lib.sshash_debug_counters(out8, 1)
    print("member rep", rep, "bad", int((mo == 0).sum().item()), "checked", out8[0], "LDS != global", out8[1], "of which final miss", out8[2])
