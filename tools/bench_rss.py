"""Peak host memory and wall time of one bench.py run (the child's ru_maxrss): python tools/bench_rss.py [bench.py arguments]"""
import os, resource, subprocess, sys, time
t0 = time.time()
rc = subprocess.call([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py")] + sys.argv[1:])
ru = resource.getrusage(resource.RUSAGE_CHILDREN)
print("rc %d  wall %.1f s  peak host memory of the largest child %.2f GiB  user %.0f s  sys %.0f s" % (rc, time.time() - t0, ru.ru_maxrss / 1048576.0, ru.ru_utime, ru.ru_stime), file=sys.stderr)
sys.exit(rc)
