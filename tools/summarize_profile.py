#!/usr/bin/env python3
"""Condense rocprofv3 output (rocpd sqlite: kernel stats + FETCH_SIZE / WRITE_SIZE counter passes) into markdown.

usage: summarize_profile.py <dir with stats/ fetch/ write/ sub-directories>
HBM traffic follows MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts a wide
coalesced streaming read at exactly half its bytes, so the read side is doubled ("corrected" column).
"""
import glob
import os
import sqlite3
import sys


def db_of(d):
    hits = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    return sqlite3.connect(hits[0]) if hits else None


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0][:70]


def main(out):
    traffic = {}
    print("# rocprofv3 summary: %s\n" % os.path.basename(os.path.normpath(out)))
    db = db_of(os.path.join(out, "stats"))
    if db:
        print("## kernel stats (`rocprofv3 --kernel-trace --stats`), grouped by kernel and grid\n")
        print("| kernel | grid.x | calls | total ms | avg us |\n|---|---|---|---|---|")
        q = "select name, grid_x, count(*), sum(duration), avg(duration) from kernels group by name, grid_x order by sum(duration) desc limit 24"
        for name, gx, n, tot, avg in db.execute(q):
            print("| %s | %d | %d | %.3f | %.2f |" % (short(name), gx, n, tot / 1e6, avg / 1e3))
            traffic.setdefault("%s@%d" % (short(name), gx), {})["avg_us"] = avg / 1e3
    for sub, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        db = db_of(os.path.join(out, sub))
        if not db:
            continue
        print("\n## %s per dispatch (`rocprofv3 --pmc %s --kernel-trace`, its own pass)\n" % (counter, counter))
        print("| kernel | grid.x | dispatches | mean %s (KiB) | MB per dispatch%s |\n|---|---|---|---|---|"
              % (counter, ", read side x2 (gfx950 correction)" if counter == "FETCH_SIZE" else ""))
        q = ("select kernel_name, grid_size_x, count(*), avg(value) from counters_collection where counter_name = ? "
             "group by kernel_name, grid_size_x order by sum(value) desc limit 64")
        for name, gx, n, v in db.execute(q, (counter,)):
            mb = v * 1024 / 1e6 * (2 if counter == "FETCH_SIZE" else 1)
            print("| %s | %d | %d | %.1f | %.2f |" % (short(name), gx, n, v, mb))
            traffic.setdefault("%s@%d" % (short(name), gx), {})["fetch_bytes_x2" if counter == "FETCH_SIZE" else "write_bytes"] = mb * 1e6
    import json
    with open(os.path.join(out, "traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1])
