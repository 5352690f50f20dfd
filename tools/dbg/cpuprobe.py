import os, time, torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
print("loadavg", open("/proc/loadavg").read().strip())
a = torch.randn(2048, 2048); b = torch.randn(2048, 2048)
for n in (4, 8, 16, 32, 64, 128):
    torch.set_num_threads(n)
    (a @ b)
    t = time.time()
    for _ in range(5): (a @ b)
    dt = (time.time() - t) / 5
    print("threads", n, "matmul 2048^3: %.1f ms  %.1f GFLOP/s" % (dt * 1e3, 2 * 2048 ** 3 / dt / 1e9), flush=True)
