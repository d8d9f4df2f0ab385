"""Summarise a rocprofv3 kernel trace CSV: per-kernel mean duration and the overlap between kernels.
usage: python tools/timeline.py <kernel_trace.csv> [first_n_rows_to_print]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("Queue_Id", "?")) for r in rows))
ev = ev[len(ev) // 2:]                               # the steady-state half
dur = collections.defaultdict(list)
for s, e, n, q in ev:
    dur[n].append((e - s) / 1e3)
for n, d in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("%-62s n=%5d mean %7.2f us  sum %9.1f us" % (n, len(d), sum(d) / len(d), sum(d)))
# union of busy time and the sum of durations
t0, t1 = ev[0][0], max(e for _, e, _, _ in ev)
busy, cur_s, cur_e = 0, None, None
for s, e, _, _ in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _, _ in ev)
print("span %.1f us, busy (union) %.1f us, sum of durations %.1f us -> mean concurrency %.2f, idle %.1f%%"
      % ((t1 - t0) / 1e3, busy / 1e3, tot / 1e3, tot / busy, 100.0 * (1 - busy / (t1 - t0))))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 0
base = ev[0][0]
for s, e, nme, q in ev[:n]:
    print("%9.2f -> %9.2f  (%6.2f)  q%s  %s" % ((s - base) / 1e3, (e - base) / 1e3, (e - s) / 1e3, q, nme[:40]))
