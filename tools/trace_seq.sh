export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 $GRAFT_REPO_ROOT/tools/run_workload.py uniform 2 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/tr/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].split('(')[0][:60] for r in rows]
# last forward: find last pack_xyzr
idx = max(i for i, n in enumerate(names) if 'pack_xyzr' in n)
prev = None; cnt = 0
for n in names[idx:]:
    if n == prev: cnt += 1; continue
    if prev: print(cnt, prev)
    prev, cnt = n, 1
print(cnt, prev)
PY
