#!/bin/bash
# 1024-thread x 64-feature strip kernel: alone, and overlapped with the gather kernel (kernel-trace timestamps)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r02_p37; rm -rf $out; mkdir -p $out
PGCN_CORE_OVERLAP=2 timeout 300 python -m pytest tests/test_hip_gpu.py -m gpu -x -q -k "strip or spmm" > $out/tests2.txt 2>&1; grep -E "passed|failed|error|^E  " $out/tests2.txt | tail -3
run() { tag=$1; shift
  env "$@" python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>$out/$tag.err > $out/$tag.json
  python -c "
import json;r=json.load(open('$out/$tag.json'));print('$tag',r['ms_per_step'],r['roofline']['avg_launch_ms'],r['roofline'].get('avg_launch_ms_backward_AT'),r['roofline']['split_us'])"
}
run half_alone PGCN_STRIP_HALF=1
run ov2 PGCN_CORE_OVERLAP=2
run ov1 PGCN_CORE_OVERLAP=1
PGCN_CORE_OVERLAP=2 rocprofv3 --kernel-trace --kernel-include-regex "spmm" --output-format csv -d $out/trace -- python tools/spmm_probe.py --once s8c1024k > $out/log.txt 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r02_p37/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[-4]['Start_Timestamp'])
for r in rows[-4:]:
    print("%-40s start %9.1f us  end %9.1f us  dur %7.1f  queue %s" % (r['Kernel_Name'][28:68], (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r.get('Queue_Id','?')))
PY
