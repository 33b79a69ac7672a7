"""Read-stream sharding across the GPUs of one node.

Every (read, subgraph) problem is independent, so the stream is cut into
contiguous blocks, one per rank (one process per GPU); there is NO data-path
collective.  torch.distributed is used only for the barrier / MAX-reduce of the
timing in bench.py and for gathering results in tests.
"""
import os


def env_rank():
    """(rank, local_rank, world_size) as torch.distributed.run exports them."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(n_total, rank, world, group=1):
    """Contiguous block [begin, end) of a stream of n_total problems for `rank`.  Cuts fall on multiples of `group` (2 keeps the
    two reads of a pair on one GPU); block sizes differ by at most `group`."""
    n_groups = (n_total + group - 1) // group
    base, extra = divmod(n_groups, world)
    begin = rank * base + min(rank, extra)
    end = begin + base + (1 if rank < extra else 0)
    return min(begin * group, n_total), min(end * group, n_total)


def usable_cpus():
    """CPUs this process may really use: the affinity mask cut by the cgroup CPU quota (a container on a 256-thread host is often
    given far fewer CPUs' worth of time; os.cpu_count() does not see that)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return n


def host_threads_per_rank(world, cores=None):
    """Packing / unpacking threads one rank may use when `world` ranks share the host (bench.py caps VGAMD_HOST_THREADS with it):
    the engine's default is min(cores, 48) per process, which 8 ranks would oversubscribe."""
    cores = cores or usable_cpus()
    return max(1, min(48, cores // max(world, 1)))


def align_shard(engine, problems, rank, world, ops_per_problem=0):
    """Align this rank's block of `problems` (a list of problem dicts) and return
    (begin, results, cigars) with results as a numpy record array."""
    from . import capi
    begin, end = shard_range(len(problems), rank, world)
    if end == begin:
        return begin, None, []
    ps = capi.ProblemSet.from_lists(problems[begin:end])
    res, ops = engine.align(ps, ops_per_problem)
    cigars = [capi.cigar_string(res[i], ops) if res["score"][i] > 0 else "" for i in range(ps.n)]
    return begin, res, cigars
