"""Collective bookkeeping of the data-parallel path (operators/distributed_wrapper.py:41-43 / DDP + SyncBatchNorm in
the reference issue their collectives implicitly; here every one is an explicit call and is counted).

Two communicators are in use when world_size > 1: the default one (SyncBN statistic exchanges, enqueued in program
order from the forward / backward nodes) and the gradient buckets' own one (rrnet_amd.flat, launched from the backward
nodes the moment a bucket is complete).  RCCL requires that every rank issues the SAME sequence of collectives on each
communicator; a rank-dependent sequence (a layer skipped on one rank, a bucket launched in a different order) is a
hang on real hardware, not a wrong number.  `RR_DP_TRACE=1` records, per rank, the sequence of
(communicator, op, element count) of everything issued plus the `mark_ready` reports of the parameters, so that a
2-rank test on one box can assert what would otherwise only show up as a hang on an 8-GPU node
(tests/test_dp_gpu.py::test_two_rank_collective_sequences_are_identical).  The counters are always on (two integer
adds per collective): bench.py reports `collectives_per_step`."""
import os

ENABLED = os.environ.get("RR_DP_TRACE", "0") == "1"
# RR_DP_FORCE=1: a process group of ONE rank is treated as data parallel — SyncBN statistic exchanges, parameter / buffer
# broadcasts and the bucketed gradient all-reduce all go through the backend (RCCL on the GPU box) as identity
# collectives.  A 1-GPU box cannot host two RCCL ranks; this is how the real backend's stream semantics (collectives
# ordered behind the CURRENT stream, async work handles, a second communicator used from autograd worker threads) meet
# the code that was written for them (tests/test_dp_gpu.py::test_rccl_single_rank_*; bench.py under RR_DP_FORCE=1).
FORCE = os.environ.get("RR_DP_FORCE", "0") == "1"


def dp_active():
    """True when collectives must be issued: a process group of more than one rank (or of one rank under RR_DP_FORCE)."""
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE)
EVENTS = []                 # (communicator, op, numel, note)
COUNTS = {}                 # communicator -> collectives issued since the last reset()


def record(comm, op, numel, note=""):
    COUNTS[comm] = COUNTS.get(comm, 0) + 1
    if ENABLED:
        EVENTS.append((comm, op, int(numel), note))


HOST_S = {}                 # communicator -> host seconds spent inside its collective calls since the last reset()


def all_reduce(t, group, comm, note=""):
    """A blocking-semantics all-reduce (ordered behind, and ahead of, the current stream) of the data-parallel path: counted,
    traced, and its HOST time accumulated per communicator — bench.py reports it as `dp_host_ms_per_step`, the part of the
    forced one-rank step's overhead that is enqueue cost rather than device idle around the exchange."""
    import time
    import torch.distributed as dist
    record(comm, "all_reduce", t.numel(), note)
    h0 = time.perf_counter()
    dist.all_reduce(t, group=group)
    HOST_S[comm] = HOST_S.get(comm, 0.0) + time.perf_counter() - h0


def mark(note):
    """A non-collective event whose position in the sequence matters (a parameter's gradient reported complete)."""
    if ENABLED:
        EVENTS.append(("-", "mark", 0, note))


def reset():
    EVENTS.clear()
    COUNTS.clear()
    HOST_S.clear()


def counts():
    return dict(COUNTS)


def sequence(comm=None):
    """The recorded collectives (marks excluded) of one communicator, or of all in issue order."""
    return [e for e in EVENTS if e[1] != "mark" and (comm is None or e[0] == comm)]
