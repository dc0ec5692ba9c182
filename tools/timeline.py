"""Summarise a rocprofv3 kernel trace: median / p90 duration per kernel over the LAST decode."""
import csv, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if r['Kernel_Name'].startswith(('k_', 'void k_', 'void jd_gmm', 'jd_'))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last third of the trace = last decode of three
rows = rows[2 * len(rows) // 3:]
by = {}
for r in rows:
    n = r['Kernel_Name'].split('(')[0].replace('void ', '')
    by.setdefault(n, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000.0)
tot = 0.0
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    v = np.array(v); tot += v.sum()
    print("%-22s n=%5d sum=%8.1f ms  mean=%7.1f  median=%7.1f  p90=%7.1f  max=%8.1f us" % (n, len(v), v.sum() / 1e3, v.mean(), np.median(v), np.percentile(v, 90), v.max()))
print("total kernel time %.1f ms; span %.1f ms" % (tot / 1e3, (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e6))
