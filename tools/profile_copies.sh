out=gpurun_out/r03_copies
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $out/trace -o trace -- python3 bench.py --steps 3 --warmup 2 --no-decode --no-cpu-baseline > $out/trace.log 2>&1
python3 - <<'PY'
import sqlite3, glob, collections
db = sqlite3.connect(glob.glob("gpurun_out/r03_copies/trace/*.db")[0])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
print(cols)
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = db.execute(f"select d.start, d.end, s.kernel_name, d.{qcol} from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
marks = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
seg = rows[marks[-2] + 1:marks[-1] + 1]
byq = collections.Counter((r[3], r[2][:40]) for r in seg if "copyBuffer" in r[2] or "FillFunctor" in r[2] or "fillBuffer" in r[2] or "direct_copy" in r[2])
for k, v in sorted(byq.items(), key=lambda x: -x[1])[:12]: print(v, k)
qs = collections.Counter(r[3] for r in seg); print("launches per queue", qs)
# what precedes / follows the copyBuffer kernels on their queue
prev = collections.Counter(); nxt = collections.Counter()
byqueue = collections.defaultdict(list)
for r in seg: byqueue[r[3]].append(r)
for q, lst in byqueue.items():
    for i, r in enumerate(lst):
        if "copyBuffer" in r[2]:
            if i > 0: prev[lst[i-1][2][:60]] += 1
            if i + 1 < len(lst): nxt[lst[i+1][2][:60]] += 1
print("before copyBuffer:", prev.most_common(6)); print("after copyBuffer:", nxt.most_common(6))
PY
rm -rf $out/trace
