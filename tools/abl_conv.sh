cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats -d gpurun_out/_cv -o cv -- python3 tools/bench_conv.py > /dev/null 2>&1
python3 - <<'PY'
import sqlite3,glob
db=sqlite3.connect(glob.glob('gpurun_out/_cv/*.db')[0]); cur=db.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
rows=cur.execute(f"select s.kernel_name, d.end-d.start from {kd} d join {ks} s on d.kernel_id=s.id").fetchall()
import collections
agg=collections.defaultdict(list)
for n,d in rows:
    if 'k_conv' in n: agg[n[:60]].append(d)
for n,v in agg.items():
    v=sorted(v); print("%-60s n=%3d median %8.1f us  max %8.1f"%(n,len(v),v[len(v)//2]/1e3,v[-1]/1e3))
PY
rm -rf gpurun_out/_cv
