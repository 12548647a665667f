"""Where does the host CPU of a bench run go?  Runs bench.py in-process and samples /proc/self/task/*/stat:
CPU seconds per thread class (executor workers = the request threads; main; other long-lived = HIP runtime;
short-lived = per-call C++ threads: layout analysis, unpacking).  Usage: python tools/cpu_sampler.py [bench flags]"""
import os
import sys
import threading
import time

tck = os.sysconf("SC_CLK_TCK")
seen = {}      # tid -> (first_seen, last_cpu, comm)
stop = False


def sample():
    while not stop:
        now = time.time()
        try:
            for t in os.listdir("/proc/self/task"):
                try:
                    f = open("/proc/self/task/%s/stat" % t).read()
                    rest = f[f.rindex(")") + 2:].split()
                    cpu = (int(rest[11]) + int(rest[12])) / tck
                    if t not in seen:
                        seen[t] = [now, cpu, now]
                    else:
                        seen[t][1] = cpu
                        seen[t][2] = now
                except Exception:
                    pass
        except Exception:
            pass
        time.sleep(0.02)


th = threading.Thread(target=sample, daemon=True)
th.start()
sys.argv = ["bench.py"] + sys.argv[1:]
import runpy  # noqa: E402

t0 = time.time()
try:
    runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
finally:
    stop = True
    th.join()
    total = time.time() - t0
    long_lived = [(v[1], v[2] - v[0], t) for t, v in seen.items() if v[2] - v[0] > 1.0]
    short = [(v[1], v[2] - v[0], t) for t, v in seen.items() if v[2] - v[0] <= 1.0]
    long_lived.sort(reverse=True)
    print("wall %.1f s; threads seen %d" % (total, len(seen)), file=sys.stderr)
    print("long-lived threads (cpu s, lifetime s):", [(round(a, 2), round(b, 1)) for a, b, _ in long_lived[:16]], file=sys.stderr)
    print("short-lived threads: n=%d, cpu total %.2f s (last sampled values; lower bound)" % (len(short), sum(a for a, _, _ in short)),
          file=sys.stderr)
