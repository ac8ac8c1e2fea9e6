# GAT: Z and both attention projections as ONE product (PGAT.forward, gat.GatAggregatePacked): parity tests, then the epoch
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r05_p13; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_gat_gpu.py -m gpu -q -x > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
for rep in 1 2; do
python bench.py --workload reddit-gat --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_gat_$rep.json 2> $out/bench_gat_$rep.err
python -c "
import json; r=json.load(open('$out/bench_gat_$rep.json')); print('GAT ms/epoch %.2f'%r['ms_per_step'], 'dominant pass %.3f ms'%r['roofline']['avg_launch_ms'], 'loss', r.get('loss'))" || tail -5 $out/bench_gat_$rep.err
done
python bench.py --workload reddit-gat --emulate-rank 0/4 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_gat_rank_0_4.json 2> $out/bench_gat_rank.err; python -c "
import json; r=json.load(open('$out/bench_gat_rank_0_4.json')); print('GAT rank 0/4 ms/epoch %.2f'%r['ms_per_step'])"
