#!/bin/bash
# Round 5, GPU call 5: fp32 kernel with plain dwordx2 rows (GL_ROWQUAD=0) against the product build; smoke(); soak; path fuzz; a 2-rank gloo rehearsal
OUT=gpurun_out/r05c5; mkdir -p $OUT
timeout -k 10 300 python3 scripts/cmp_builds.py abtest/r05/lib_fin.so abtest/r05/lib_q0.so > $OUT/cmp_q0.txt 2>&1; cat $OUT/cmp_q0.txt
for g in T D S; do GOMEL_RUNS=1 GOMEL_FP64_ITERS=0 timeout -k 10 600 python3 scripts/ab_time.py 7 $g abtest/r05/lib_fin.so abtest/r05/lib_q0.so > $OUT/ab_fp32_rowquad_$g.txt 2>&1; echo "fp32 $g"; cat $OUT/ab_fp32_rowquad_$g.txt; done
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -3 $OUT/smoke.txt
timeout -k 10 300 python3 scripts/soak.py 60 > $OUT/soak.txt 2>&1; tail -3 $OUT/soak.txt
timeout -k 10 400 python3 scripts/fuzz_paths.py 300 5 > $OUT/fuzz_paths.txt 2>&1; tail -3 $OUT/fuzz_paths.txt
GOMEL_BENCH_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 2 --utts-per-gpu 256 --steps 3 --warmup 1 > $OUT/bench_2rank_gloo_one_gpu.json 2> $OUT/bench_2rank_gloo_one_gpu.err; tail -c 600 $OUT/bench_2rank_gloo_one_gpu.json
