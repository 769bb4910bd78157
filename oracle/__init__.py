"""oracle/ -- CPU restatement of the reference's Lookup path and an independent ground truth.

TEST INFRASTRUCTURE ONLY: nothing under sshash_amd/ or include/ may import, link or call into
this package. Allowed users: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
"""
