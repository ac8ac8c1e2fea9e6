#!/bin/bash
# r03 batch 6: full GPU test suite after the pruning refactor; shard shapes with the degree-mass round cut and the
# adaptive strip pieces; kernel trace of one emulated rank
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r03_p6; rm -rf $out; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1; tail -3 $out/pytest_gpu.txt
run() {  # tag, rank spec, tuning
  tag=$1; rp=$2; tun=$3
  PGCN_TUNING="$tun" python bench.py --emulate-rank $rp --steps 10 --warmup 2 --no-cpu-baseline > $out/b_$tag.json 2> $out/b_$tag.err
  python - <<PY
import json
try:
    r=json.load(open("$out/b_$tag.json"))
    h=r.get("halo_groups") or []
    print("%-22s ms/epoch %.3f  A_loc %.3f ms %s | halo %s nnz %s | n_halo %d" % ("$tag", r["ms_per_step"], r["roofline"]["avg_launch_ms"], {k: round(v) for k, v in (r["roofline"].get("split_us") or {}).items()}, ["%.3f" % x["avg_launch_ms"] for x in h], [x["nnz"] for x in h], r["config"]["rank_shape"]["n_halo"]))
except Exception as e:
    print("$tag failed", e)
PY
}
for rp in 0/8 0/4 0/2; do
  t=$(echo $rp | tr '/' '_')
  run new_$t $rp ""
  run m0_$t $rp "round_mass_permille=0"
  run m500_$t $rp "round_mass_permille=500"
  run m850_$t $rp "round_mass_permille=850"
  run r1_$t $rp "exchange_rounds=1"
done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_n1.json 2> $out/bench_n1.err; python -c "
import json; r=json.load(open('$out/bench_n1.json')); print('N=1 ms/epoch', r['ms_per_step'], 'spmm', r['roofline']['avg_launch_ms'], r['roofline']['split_us'])"
rocprofv3 --kernel-trace --output-format csv -d $out/trace -o r8 -- python bench.py --emulate-rank 0/8 --steps 2 --warmup 1 --no-cpu-baseline > $out/trace_stdout.log 2> $out/trace_stderr.log
python - <<PY
import csv, glob
f = glob.glob("$out/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last 140 kernels = the last epoch or so
last = rows[-140:]
t0 = int(last[0]["Start_Timestamp"])
with open("$out/trace_tail.txt", "w") as fh:
    prev_end = t0
    for r in last:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        fh.write("%9.1f us  gap %6.1f  dur %7.1f  %s\n" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:90]))
        prev_end = e
PY
rm -rf $out/trace
head -70 $out/trace_tail.txt
