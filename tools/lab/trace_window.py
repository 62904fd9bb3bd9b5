"""Lab: one batch of the calibrated native-executor stream out of a rocprofv3 kernel trace (rocpd sqlite): kernels in order with their durations."""
import sqlite3, sys
from collections import Counter
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute("select s.display_name, d.start, d.end, d.grid_size_x from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id order by d.start"))
names = [r[0] for r in rows]
pl = [i for i, n in enumerate(names) if 'pair_ll32h' in n]
wins = {}
for k in range(len(pl) - 1):
    a, b = pl[k], pl[k + 1]
    nl = sum(1 for i in range(a, b) if 'lstm' in names[i].lower())
    nt = sum(1 for i in range(a, b) if 'at::native' in names[i])
    wins.setdefault((nl, nt, b - a), []).append((a, b))
key = min((k for k in wins if k[0] > 0), key=lambda k: k[1])
a, b = wins[key][len(wins[key]) // 2]
print(key, len(wins[key]), "windows")
tot = Counter()
for i in range(a, b):
    n, s, e, g = rows[i]
    short = n.replace('(anonymous namespace)::', '').replace('void ', '')[:44]
    print("%-44s %8.1f us" % (short, (e - s) / 1000))
    tot[short.split('(')[0].split('<')[0]] += (e - s) / 1000
print("span us", (rows[b][1] - rows[a][1]) / 1000)
print({k: round(v, 1) for k, v in tot.most_common()})
