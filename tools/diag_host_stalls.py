"""Host-side scheduling stalls on the box: spin on the clock for a few seconds and report every
gap above 2 ms between two consecutive reads (and /sys/fs/cgroup/cpu.stat throttling counters)."""
import time


def cpu_stat():
    try:
        with open("/sys/fs/cgroup/cpu.stat") as f:
            return dict(line.split() for line in f)
    except OSError:
        return {}


before = cpu_stat()
t_end = time.perf_counter() + 8.0
last = time.perf_counter()
gaps = []
n = 0
while True:
    now = time.perf_counter()
    if now - last > 2e-3:
        gaps.append((round(now - t_end + 8.0, 3), round(1e3 * (now - last), 2)))
    last = now
    n += 1
    if now > t_end:
        break
after = cpu_stat()
print("reads:", n, "gaps > 2 ms (t, ms):", gaps[:40])
for k in ("nr_periods", "nr_throttled", "throttled_usec"):
    if k in before:
        print(k, int(after[k]) - int(before[k]))
