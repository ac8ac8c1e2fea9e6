#!/bin/bash
# do the strip (side stream) and gather (main stream) kernels really run at the same time?  kernel-trace timestamps
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r02_p36; rm -rf $out; mkdir -p $out
PGCN_CORE_OVERLAP=2 rocprofv3 --kernel-trace --kernel-include-regex "spmm" --output-format csv -d $out/trace -- python tools/spmm_probe.py --once s8c1024k > $out/log.txt 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r02_p36/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
for r in rows[-8:]:
    print("%-40s start %9.1f us  end %9.1f us  dur %7.1f  queue %s" % (r['Kernel_Name'][28:68], (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r.get('Queue_Id','?')))
PY
