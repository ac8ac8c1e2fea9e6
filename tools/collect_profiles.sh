#!/bin/bash
# Copy the summaries of gpurun_out/final_<tag> (scratch) into profiles/ (tracked).  usage: bash tools/collect_profiles.sh r03
tag=${1:-r03}; src=gpurun_out/final_$tag; dst=profiles
cp $src/pytest_gpu_full.txt $dst/${tag}_pytest_gpu.txt 2>/dev/null || cp $src/pytest_gpu.txt $dst/${tag}_pytest_gpu.txt
cp $src/bench.json $dst/${tag}_bench_stdout.json
cp $src/pmc_traffic.json $dst/pmc_traffic.json
for w in products sbm mid gat gat_noblocks; do [ -s $src/bench_$w.json ] && cp $src/bench_$w.json $dst/${tag}_bench_${w}_stdout.json; done
for f in $src/bench_rank_*.json $src/bench_gat_rank_*.json $src/bench_products*_sbm_*.json $src/bench_papers_full_rank_*.json $src/papers_full_rank_0_8_check.json; do [ -s "$f" ] && cp "$f" $dst/${tag}_$(basename $f); done
for f in $src/pmc_summary_*.txt; do n=$(basename $f .txt | sed 's/pmc_summary_//'); grep -v "^$" $f > $dst/${tag}_pmc_$n.txt; done
st=$(ls $src/prof/*kernel_stats.csv $src/prof/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$st" ] && head -45 "$st" > $dst/${tag}_bench_kernel_stats.csv
sg=$(ls $src/prof_gat/*kernel_stats.csv $src/prof_gat/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$sg" ] && grep -E "Name|spmm_heads|gat_|csr_row_sums|fixup|nll|rows_kernel|split_panels" "$sg" | head -30 > $dst/${tag}_bench_gat_kernel_stats.csv
[ -f gpurun_out/parity_observed.jsonl ] && cp gpurun_out/parity_observed.jsonl $dst/${tag}_parity_observed.jsonl
ls -la $dst | grep ${tag}_
