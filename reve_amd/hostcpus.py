"""How many CPUs the process may really use (bench.py's CPU baseline, the tests' oracle threads; the C++ side has the same
logic in dirmode.cpp effective_cpus)."""
import os


def usable_cpus(root: str = ""):
    """CPUs this process may use: its affinity mask, cut down to its control group's CPU quota (cgroup v2 `cpu.max`, v1
    `cpu.cfs_quota_us`).  The GPU boxes of this project show 128-256 CPUs and allow a process 16 of them per 100 ms period: an
    OpenMP team of 256 threads there is throttled as a whole and runs no faster than 16."""
    n = len(os.sched_getaffinity(0))
    quotas = []
    def cpu_max(path):
        try:
            a, b = open(path).read().split()[:2]
            if a != "max" and int(b) > 0:
                quotas.append(int(a) / int(b))
        except (OSError, ValueError):
            pass
    cpu_max(root + "/sys/fs/cgroup/cpu.max")
    try:
        for l in open(root + "/proc/self/cgroup").read().splitlines():
            if l.startswith("0::"):
                path = l[3:]
                while len(path) > 1:
                    cpu_max(root + "/sys/fs/cgroup" + path + "/cpu.max")
                    path = path.rsplit("/", 1)[0]
    except OSError:
        pass
    for d in (root + "/sys/fs/cgroup/cpu", root + "/sys/fs/cgroup/cpu,cpuacct"):
        try:
            q, per = int(open(d + "/cpu.cfs_quota_us").read()), int(open(d + "/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                quotas.append(q / per)
        except (OSError, ValueError):
            pass
    q = min(quotas) if quotas else n
    return max(1, min(n, int(q))), n
