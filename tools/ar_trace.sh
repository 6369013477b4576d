set -x
out=gpurun_out/r04_ar_trace
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $out/t -o t -- python3 tools/ar_step_probe.py > $out/run.log 2>&1
python3 - <<'PY'
import sqlite3, glob, collections
db = sqlite3.connect(glob.glob("gpurun_out/r04_ar_trace/t/*.db")[0])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "info_kernel_symbol" in t][0]
rows = db.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
# last frame: from the last ar_embed kernel to the end
idx = [i for i, r in enumerate(rows) if "ar_embed" in r[2]]
seg = rows[idx[-2]:idx[-1]]
print("launches in one frame:", len(seg), "span us:", (seg[-1][1] - seg[0][0]) / 1e3)
busy = sum(r[1] - r[0] for r in seg) / 1e3
print("sum of kernel durations us:", busy)
agg = collections.OrderedDict()
for a, b, n in seg:
    k = n.split("(")[0][-40:]
    v = agg.setdefault(k, [0, 0.0]); v[0] += 1; v[1] += (b - a) / 1e3
for k, (c, us) in agg.items():
    print(f"{k:42s} x{c:3d} {us / c:7.2f} us each")
gaps = [(seg[i + 1][0] - seg[i][1]) / 1e3 for i in range(len(seg) - 1)]
print("gaps us: mean %.2f min %.2f max %.2f" % (sum(gaps) / len(gaps), min(gaps), max(gaps)))
PY
rm -rf $out/t
