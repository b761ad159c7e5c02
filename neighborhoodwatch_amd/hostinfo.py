"""How many CPUs this process may really use.  On the GPU boxes ``os.cpu_count()`` and the affinity mask say 256 while the container's cgroup
allows 16 CPUs' worth of time: thread pools sized by the former (pyarrow's, BLAS's) spend the quota on contention -- parquet decode costs
0.55 CPU-seconds per GB with pyarrow's pool at 256 threads and 0.40 at 16 (tools/arrow_pool_probe.py, profiles/r05at_arrow_pool.log), and CPU
seconds are what the file path is bound by.  No counterpart in the reference (it leaves every pool at its default)."""
import math
import os


def effective_cpus() -> int:
    """min(affinity mask, cgroup CPU quota rounded up); at least 1."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:                                                        # cgroup v2: "<quota> <period>" or "max <period>"
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(p)
    except (OSError, ValueError):
        try:                                                    # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, math.ceil(quota)))
    return max(1, n)


_arrow_sized = False


def size_arrow_pool() -> int:
    """Once per process: shrink pyarrow's CPU pool to ``effective_cpus()`` when it was sized by a larger machine-wide count (never grows it;
    ``NWKNN_ARROW_THREADS=<n>`` sets it outright, ``0`` leaves pyarrow alone).  -> the pool size in force."""
    global _arrow_sized
    import pyarrow as pa
    if not _arrow_sized:
        _arrow_sized = True
        want = os.environ.get("NWKNN_ARROW_THREADS")
        if want is not None:
            if int(want) > 0:
                pa.set_cpu_count(int(want))
        elif pa.cpu_count() > effective_cpus():
            pa.set_cpu_count(effective_cpus())
    return pa.cpu_count()
