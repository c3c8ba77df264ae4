"""Per-kernel durations and the gaps of a toy-MC call (bi_eval_datasets, C3) from a rocprofv3 --kernel-trace database:
    rocprofv3 --kernel-trace --memory-copy-trace -d OUT -o c3 -- python3 bench.py --config C3 --no-cpu-baseline --steps 10
    python tools/profile/c3_call_timeline.py OUT/c3_results.db"""
import sqlite3, statistics, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
names = ('k_morph_logmu', 'k_dataset_dot_tiled', 'k_dataset_finish')
calls = []
for i in range(len(rows) - 2):
    if all(n in rows[i + k][0] for k, n in enumerate(names)):
        calls.append(rows[i:i + 3])
calls = calls[2:]                      # warm-up
for k, n in enumerate(names):
    d = [c[k][2] - c[k][1] for c in calls]
    print('%-22s median %6.1f us  (min %6.1f, %d calls)' % (n, statistics.median(d) / 1e3, min(d) / 1e3, len(d)))
span = [c[2][2] - c[0][1] for c in calls]
period = [b[0][1] - a[0][1] for a, b in zip(calls, calls[1:])]
print('first kernel start -> last kernel end: median %.1f us; call to call: median %.1f us' % (
    statistics.median(span) / 1e3, statistics.median(period) / 1e3))
