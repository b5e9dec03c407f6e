"""Busy time versus wall time of the kernels in a rocprofv3 kernel_trace.csv: python3 tools/trace_gaps.py FILE [FRACTION [END]]
looks at the launches between 1 - FRACTION and END (default 1.0) of the trace, counted in launches."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
stop = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rows = rows[int(len(rows) * (1 - frac)):int(len(rows) * stop)]
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
busy, per, cnt = 0, defaultdict(int), defaultdict(int)
end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += max(0, e - max(s, end))
    end = max(end, e)
    per[r["Kernel_Name"][:70]] += e - s
    cnt[r["Kernel_Name"][:70]] += 1
print(f"wall {1e-6*(t1-t0):.3f} ms  busy {1e-6*busy:.3f} ms  idle {100*(1-busy/(t1-t0)):.1f} %  launches {len(rows)}")
for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:25]:
    print(f"{1e-6*v:9.3f} ms {100*v/(t1-t0):5.1f} %  x{cnt[k]:5d}  {k}")
