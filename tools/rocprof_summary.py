#!/usr/bin/env python
"""Summarise rocprofv3 result databases (rocpd sqlite) into the small text files committed under profiles/.

  rocprof_summary.py stats  <results.db> <out.csv>      per-kernel Calls / total / average / min / max (like --stats csv)
  rocprof_summary.py pmc    <fetch.db> <write.db> <out.json>
        per-kernel and per-step HBM-side bytes from FETCH_SIZE / WRITE_SIZE (separate passes).  Units and the gfx950
        correction follow /opt/skills/guides/MI355X_MICROARCH.md: both counters are in KiB-like units of 1024 B as rocprofv3
        reports them here (checked against the fused AdamW kernel, whose traffic is 16 B read + 14 B written per parameter);
        FETCH_SIZE is DOUBLED (gfx950 tallies 128-B requests at 64 B).
  rocprof_summary.py gaps   <trace.db> <out.json>
        one training step (optimizer kernel to optimizer kernel) of a --kernel-trace run: wall time, time with at least one
        kernel running (union of the dispatch intervals), with >= 2 running, and the idle gaps (count, total, histogram) --
        the bubbles between dependent launches that no kernel tuning removes.
  rocprof_summary.py sq     <sq.db> <out.csv> [mark]
        per-kernel sums of every counter of one multi-counter SQ pass (SQ_WAVE_CYCLES, SQ_WAIT_ANY, SQ_ACTIVE_INST_VALU ...)
        over one unit of work (training step by default; mark = rvq_kernelI6bf16_t for one decode) + each as a share of
        SQ_WAVE_CYCLES when that counter is in the pass.
"""
import collections
import json
import re
import sqlite3
import subprocess
import sys


def _tables(db):
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    pick = lambda key: [t for t in tabs if key in t][0]
    return pick("kernel_dispatch"), pick("kernel_symbol"), ([t for t in tabs if "pmc_event" in t] or [None])[0]


def _demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(n[:-3] if n.endswith(".kd") else n for n in names),
                             capture_output=True, text=True, check=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def stats(path, out):
    db = sqlite3.connect(path)
    kd, ks, _ = _tables(db)
    rows = db.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                      f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    dm = _demangle([r[0] for r in rows])
    with open(out, "w") as f:
        f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
        for n, c, t, a, mn, mx in rows:
            f.write(f'"{dm[n]}",{c},{t},{a:.1f},{100.0 * t / total:.2f},{mn},{mx}\n')


def _per_step(path, mark="adamw_kernel"):
    db = sqlite3.connect(path)
    kd, ks, pmc = _tables(db)
    rows = db.execute(f"select d.start, s.kernel_name, d.dispatch_id, e.value from {pmc} e join {kd} d on e.event_id=d.event_id "
                      f"join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
    disp = collections.OrderedDict()
    for st, n, did, v in rows:
        disp[(st, did, n)] = disp.get((st, did, n), 0) + v
    items = list(disp.items())
    marks = [i for i, (k, _) in enumerate(items) if mark in k[2]]
    # one whole unit of work: optimizer kernel to optimizer kernel (training step) / first kernel to first kernel (one decode)
    # (decode: the 2nd and 3rd RVQ gathers of the process are two consecutive iterations of bench.py's timing loop)
    seg = items[marks[-2] + 1:marks[-1] + 1] if mark == "adamw_kernel" else items[marks[1]:marks[2]]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for (_, _, n), v in seg:
        agg[n][0] += 1; agg[n][1] += v
    return agg


def pmc(fetch_db, write_db, out):
    F, W = _per_step(fetch_db), _per_step(write_db)
    names = sorted(set(F) | set(W), key=lambda n: -(2 * F.get(n, [0, 0])[1] + W.get(n, [0, 0])[1]))
    dm = _demangle(names)
    kern = []
    for n in names:
        calls = max(F.get(n, [0, 0])[0], W.get(n, [0, 0])[0])
        rd = 2.0 * F.get(n, [0, 0])[1] * 1024.0; wr = W.get(n, [0, 0])[1] * 1024.0
        kern.append({"kernel": re.sub(r"\(anonymous namespace\)::", "", dm[n])[:160], "calls_per_step": calls,
                     "read_bytes_per_step": rd, "write_bytes_per_step": wr,
                     "bytes_per_launch": (rd + wr) / max(calls, 1)})
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --steps 2 --warmup 1`; one training "
                     "step; FETCH_SIZE doubled (gfx950), unit 1024 B", 
           "step_read_bytes": sum(k["read_bytes_per_step"] for k in kern), "step_write_bytes": sum(k["write_bytes_per_step"] for k in kern),
           "kernels": kern[:40]}
    res["step_bytes"] = res["step_read_bytes"] + res["step_write_bytes"]
    with open(out, "w") as f:
        json.dump(res, f, indent=1)


def pmc_decode(fetch_db, write_db, out):
    """HBM-side bytes of ONE Encodec decode (`bench.py --only-decode`): the dispatches from one RVQ gather to the next."""
    # the headline decode is the f32-class one (split storage): its RVQ gather is rvq_x2_kernel
    F, W = _per_step(fetch_db, "rvq_x2_kernel"), _per_step(write_db, "rvq_x2_kernel")
    names = sorted(set(F) | set(W), key=lambda n: -(2 * F.get(n, [0, 0])[1] + W.get(n, [0, 0])[1]))
    dm = _demangle(names)
    kern = []
    for n in names:
        calls = max(F.get(n, [0, 0])[0], W.get(n, [0, 0])[0])
        rd = 2.0 * F.get(n, [0, 0])[1] * 1024.0; wr = W.get(n, [0, 0])[1] * 1024.0
        kern.append({"kernel": re.sub(r"\(anonymous namespace\)::", "", dm[n])[:160], "calls_per_decode": calls,
                     "read_bytes": rd, "write_bytes": wr})
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --only-decode`; one decode of "
                     "64 x 1024 frames (f32-class, split storage); FETCH_SIZE doubled (gfx950), unit 1024 B",
           "decode_read_bytes": sum(k["read_bytes"] for k in kern), "decode_write_bytes": sum(k["write_bytes"] for k in kern),
           "kernels": kern[:24]}
    res["decode_bytes"] = res["decode_read_bytes"] + res["decode_write_bytes"]
    with open(out, "w") as f:
        json.dump(res, f, indent=1)


def gaps(path, out):
    db = sqlite3.connect(path)
    kd, ks, _ = _tables(db)
    rows = db.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
    marks = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
    seg = rows[marks[-2] + 1:marks[-1] + 1]
    t0, t1 = seg[0][0], max(r[1] for r in seg)
    ev = sorted([(r[0], 1) for r in seg] + [(r[1], -1) for r in seg])
    busy1 = busy2 = 0; depth = 0; last = t0; idle = []
    for t, dlt in ev:
        if depth >= 1: busy1 += t - last
        if depth >= 2: busy2 += t - last
        if depth == 0 and t > last: idle.append(t - last)
        depth += dlt; last = t
    hist = collections.Counter()
    for g in idle:
        hist["<1us" if g < 1000 else "1-2us" if g < 2000 else "2-5us" if g < 5000 else "5-10us" if g < 10000 else ">=10us"] += 1
    res = {"source": "rocprofv3 --kernel-trace of bench.py; one training step, optimizer kernel to optimizer kernel",
           "launches": len(seg), "wall_ms": (t1 - t0) / 1e6, "some_kernel_running_ms": busy1 / 1e6, "two_or_more_running_ms": busy2 / 1e6,
           "idle_ms": sum(idle) / 1e6, "idle_gaps": len(idle), "idle_gap_histogram": dict(hist),
           "sum_of_kernel_durations_ms": sum(r[1] - r[0] for r in seg) / 1e6}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


def sq(path, out, mark="adamw_kernel"):
    db = sqlite3.connect(path)
    kd, ks, pmc = _tables(db)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    info = [t for t in tabs if "info_pmc" in t][0]
    rows = db.execute(f"select d.start, s.kernel_name, d.dispatch_id, i.name, e.value from {pmc} e join {kd} d on e.event_id=d.event_id "
                      f"join {ks} s on d.kernel_id=s.id join {info} i on e.pmc_id=i.id order by d.start").fetchall()
    disp = collections.OrderedDict()
    for st, n, did, c, v in rows:
        disp.setdefault((st, did, n), collections.defaultdict(float))[c] += v
    items = list(disp.items())
    marks = [i for i, (k, _) in enumerate(items) if mark in k[2]]
    seg = items[marks[-2] + 1:marks[-1] + 1] if mark == "adamw_kernel" else items[marks[1]:marks[2]]
    agg, calls = collections.defaultdict(lambda: collections.defaultdict(float)), collections.Counter()
    for (_, _, n), cs in seg:
        calls[n] += 1
        for c, v in cs.items():
            agg[n][c] += v
    ctrs = sorted({c for a in agg.values() for c in a})
    base = "SQ_WAVE_CYCLES" if "SQ_WAVE_CYCLES" in ctrs else None
    names = sorted(agg, key=lambda n: -agg[n].get(base or ctrs[0], 0))
    dm = _demangle(names)
    with open(out, "w") as f:
        f.write('"Name","Calls",' + ",".join(f'"{c}"' for c in ctrs) + ("," + ",".join(f'"{c}/WAVE_CYCLES"' for c in ctrs if c != base) if base else "") + "\n")
        for n in names:
            nm = re.sub(r"\(anonymous namespace\)::", "", dm[n])[:120]
            line = f'"{nm}",{calls[n]},' + ",".join(f"{agg[n].get(c, 0):.0f}" for c in ctrs)
            if base and agg[n].get(base, 0) > 0:
                line += "," + ",".join(f"{agg[n].get(c, 0) / agg[n][base]:.3f}" for c in ctrs if c != base)
            f.write(line + "\n")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "pmc":
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
    elif sys.argv[1] == "gaps":
        gaps(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "sq":
        sq(*sys.argv[2:5])
    elif sys.argv[1] == "pmc_decode":
        pmc_decode(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        raise SystemExit(__doc__)
