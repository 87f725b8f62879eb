set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/tv; mkdir -p $OUT; cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_solver.py -m gpu -x -q -k "tv or TV" > $OUT/tests.log 2>&1; echo "rc=$?"; tail -5 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/p1 -- python3 $R/tools/time_tv_long.py > /dev/null 2>&1
export AOADMM_TV_SEQ_LONG=1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/p2 -- python3 $R/tools/time_tv_long.py > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob, os
for d in ('p1','p2'):
    f = glob.glob(os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out/tv',d,'**/*kernel_trace.csv'), recursive=True)
    if not f: print(d,'no trace'); continue
    rows=[r for r in csv.DictReader(open(f[0])) if 'prox_tv' in r['Kernel_Name']]
    print(d, [(r['Kernel_Name'][:40], r['Workgroup_Size'] if 'Workgroup_Size' in r else '', (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for r in rows][4::5])
PY
rm -rf $OUT/p1 $OUT/p2
