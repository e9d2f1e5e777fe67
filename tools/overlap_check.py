"""kernel_trace.csv -> how much kernel time ran concurrently: sum of durations vs length of their union, per queue"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")) for r in rows)
tot = sum(e - s for s, e, _ in iv)
union, cur_s, cur_e = 0, None, None
for s, e, _ in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None: union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
q = collections.Counter(x[2] for x in iv)
print(f"kernels {len(iv)}  sum {tot/1e6:.2f} ms  union {union/1e6:.2f} ms  overlapped {100*(tot-union)/tot:.1f} %  queues {dict(q)}")
