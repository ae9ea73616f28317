"""ORACLE — test infrastructure only.

CPU restatement of the reference's hot path (plain C for Soft-NMS / hard NMS, numpy +
torch-CPU for everything else).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; the product (rrnet_amd/) never does.
Each function cites the /root/reference file:line it restates.
"""
