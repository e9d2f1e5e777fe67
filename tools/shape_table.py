"""Print the per-shape conv/GEMM table written by `bench.py --shape-profile`."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['t'] = float(r['main_ms']) + float(r['reduce_ms'])
tot = sum(r['t'] for r in rows)
print("total ms/edit", round(tot, 2))
for r in sorted(rows, key=lambda r: -r['t'])[:int(sys.argv[2]) if len(sys.argv) > 2 else 50]:
    n = float(r['launches'])
    print(f"M={r['M']:>6} N={r['N']:>5} K={r['K']:>6} c3={r['conv3']} tile={r['tile']:>3} ks={r['ksplit']:>2} n={int(n):4d} "
          f"main={float(r['main_ms']) / n * 1e3:7.1f}us red={float(r['reduce_ms']) / n * 1e3:5.1f}us tot={r['t']:6.2f}ms "
          f"{float(r['gflop']) / float(r['main_ms']):6.0f} TF/s")
