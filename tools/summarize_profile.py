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
    roofline_durations(out)


def roofline_durations(out):
    """rocprofv3 durations of the SAME launches bench.py times with HIP events: the bench line (last JSON line of stats.log) says which
    kernel its roofline leg launched last (`roofline.trace_name`, `roofline.reps`) and in which order its `kernels` rows launched their
    groups (`trace`: kernel, grid, group index, launches per group, timed launches).  -> <out>/roofline_rocprof.json, one entry keyed by
    workload / objects / batch; tools/profile_bench.sh merges it into profiles/roofline_rocprof.json, which bench.py attaches to its line."""
    import hashlib
    import json
    db = db_of(os.path.join(out, "stats"))
    log = os.path.join(out, "stats.log")
    if not db or not os.path.exists(log):
        return
    line = None
    for l in open(log):
        if l.startswith("{"):
            try:
                line = json.loads(l)
            except ValueError:
                pass
    if not line or not line.get("roofline"):
        return
    rows = [(short(n), gx, st, du) for n, gx, st, du in db.execute("select name, grid_x, start, duration from kernels order by start")]
    entry = {"bench_py_sha16": hashlib.sha256(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), "rb").read()).hexdigest()[:16],
             "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py " + " ".join(line.get("argv", [])), "kernels": {}}
    entry["head"] = "bench.py sha256[:16] " + entry["bench_py_sha16"]
    r = line["roofline"]
    if r.get("trace_name"):
        d = [du for n, gx, st, du in rows if n.startswith(r["trace_name"])][-int(r.get("reps", 1)):]
        if d:
            entry["dominant"] = {"trace_name": r["trace_name"], "launches": len(d), "avg_us": sum(d) / len(d) / 1e3, "min_us": min(d) / 1e3,
                                 "max_us": max(d) / 1e3, "hip_event_us_per_launch_same_run": r.get("us_per_launch")}
            # ... and what `--stats` prints for this kernel: the average over EVERY launch of the run with the roofline leg's grid (graph
            # replays of the timed region, the streamed leg, warm-up and the roofline leg itself)
            gx_last = [gx for n, gx, st, du in rows if n.startswith(r["trace_name"])][-1]
            every = [du for n, gx, st, du in rows if n.startswith(r["trace_name"]) and gx == gx_last]
            entry["dominant"]["all_launches"] = len(every)
            entry["dominant"]["all_launches_avg_us"] = sum(every) / len(every) / 1e3
            # replay lanes (bench.py --lanes 2, the default since round 6) run consecutive steps on two streams: a launch of the timed region
            # shares the device with the other lane's kernels and lasts longer while the steps get shorter.  So the same row again, split by
            # whether ANY other dispatch of the trace ran during the launch ("alone": what a roofline fraction is about)
            import bisect
            starts = [st for n, gx, st, du in rows]
            longest = max(du for n, gx, st, du in rows)
            alone, shared = [], []
            for i, (n, gx, st, du) in enumerate(rows):
                if not (n.startswith(r["trace_name"]) and gx == gx_last):
                    continue
                lo = bisect.bisect_left(starts, st - longest)
                hi = bisect.bisect_right(starts, st + du)
                hit = any(j != i and rows[j][2] < st + du and rows[j][2] + rows[j][3] > st for j in range(lo, hi))
                (shared if hit else alone).append(du)
            entry["dominant"]["alone_launches"], entry["dominant"]["overlapped_launches"] = len(alone), len(shared)
            if alone:
                entry["dominant"]["alone_avg_us"] = sum(alone) / len(alone) / 1e3
            if shared:
                entry["dominant"]["overlapped_avg_us"] = sum(shared) / len(shared) / 1e3
    for k in line.get("kernels", []):
        tr = k.get("trace")
        if not tr:
            continue
        d = [du for n, gx, st, du in rows if n.startswith(tr["kernel"]) and gx == tr["grid_x"]]
        g = d[tr["group"] * tr["launches"]:(tr["group"] + 1) * tr["launches"]][-tr["timed"]:]
        if len(g) == tr["timed"]:
            entry["kernels"]["%s@%d#%d" % (tr["kernel"], tr["grid_x"], tr["group"])] = {
                "row": k["kernel"], "launches": len(g), "avg_us": sum(g) / len(g) / 1e3, "hip_event_us_per_launch_same_run": k.get("us_per_launch")}
    cfg = line.get("config", {})
    key = "%s:n%d:b%d" % (cfg.get("workload_id", "north_star"), cfg.get("objects_per_scene", 0), cfg.get("batch_per_gpu", 0))
    with open(os.path.join(out, "roofline_rocprof.json"), "w") as f:
        json.dump({key: entry}, f, indent=1, sort_keys=True)
    print("\n## the roofline leg's launches (the same launches bench.py times with HIP events)\n")
    if "dominant" in entry:
        dm = entry["dominant"]
        print("%s: last %d launches, rocprofv3 avg %.1f us (min %.1f, max %.1f); HIP events in the same run: %.1f us per launch; all %d launches of "
              "the run with this grid (the `--stats` row): %.1f us - %d of them with the device to themselves: %.1f us, %d sharing it with another "
              "stream's kernels (replay lanes): %.1f us"
              % (dm["trace_name"], dm["launches"], dm["avg_us"], dm["min_us"], dm["max_us"], dm["hip_event_us_per_launch_same_run"] or 0,
                 dm.get("all_launches", 0), dm.get("all_launches_avg_us", 0.0), dm.get("alone_launches", 0), dm.get("alone_avg_us", 0.0),
                 dm.get("overlapped_launches", 0), dm.get("overlapped_avg_us", 0.0)))
    for kk, v in sorted(entry["kernels"].items()):
        print("%s  [%s]: rocprofv3 avg %.1f us over %d launches; HIP events %.1f us" % (kk, v["row"], v["avg_us"], v["launches"], v["hip_event_us_per_launch_same_run"] or 0))


if __name__ == "__main__":
    main(sys.argv[1])
