import os, time, torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(p, open(p).read().strip())
    except Exception as e: print(p, "n/a")
os.system("lscpu | egrep 'Model name|^CPU\\(s\\)|Thread|Socket|Flags' | cut -c1-400")
print("torch threads default", torch.get_num_threads())
for nt in (torch.get_num_threads(), 8, 32):
    torch.set_num_threads(nt)
    for dt in (torch.bfloat16, torch.float32):
        x = torch.randn(11, 4096).to(dt); w = torch.randn(14336, 4096).to(dt)
        torch.nn.functional.linear(x, w)
        t0 = time.time()
        for _ in range(3): torch.nn.functional.linear(x, w)
        print(f"threads {nt} {dt}: {(time.time()-t0)/3*1e3:.1f} ms per [11x4096]x[4096x14336]", flush=True)
