#!/bin/bash
# Round profile set (run on the GPU box from the repo root):  bash scripts/profile_round.sh r01
#   1. the default bench line                               -> gpurun_out/<tag>_bench.json
#   2. rocprofv3 --kernel-trace --stats of `bench.py --no-extra --no-cpu-baseline` (the resnet18 workload only, so the
#      per-kernel averages are those of the bench line's roofline kernel) -> gpurun_out/<tag>_bench_kernel_stats.csv,
#      the line it printed, and <tag>_dominant_kernel.json (rocprof average of the dominant kernel next to the bench's HIP-event average)
#   0. PMC passes FETCH_SIZE / WRITE_SIZE (separate runs)   -> gpurun_out/<tag>_pmc_{FETCH,WRITE}_SIZE_per_kernel.csv, <tag>_traffic.json
#   8. the opt-in 'f32x3p' arithmetic: bench line, kernel stats, per-layer micro timings -> <tag>_bench_f32x3p*.json, <tag>_f32x3p_kernel_stats.csv, <tag>_x3p_micro.txt
# Copy what should be judged from gpurun_out/ into profiles/.
set -o pipefail
tag=${1:-r02}
R=$(pwd)
out=$R/gpurun_out
mkdir -p $out
# 0. PMC traffic passes first (FETCH_SIZE / WRITE_SIZE in separate runs; fp32 and bf16 configurations): the bench lines
#    below look their dominant kernel's HBM bytes up in profiles/<tag>_traffic*.json, stamped with the sources' sha
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/prof_$c -- python3 $R/bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-extra --no-roofline --min-seconds 0 > $out/prof_$c.log 2>&1 || { tail -5 $out/prof_$c.log; exit 1; }
done
cd $R
python scripts/pmc_traffic.py $tag $out/prof_FETCH_SIZE $out/prof_WRITE_SIZE
rm -rf $out/prof_FETCH_SIZE $out/prof_WRITE_SIZE
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/prof16_$c -- python3 $R/bench.py --dtype bf16 --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-extra --no-roofline --min-seconds 0 > $out/prof16_$c.log 2>&1 || { tail -5 $out/prof16_$c.log; exit 1; }
done
cd $R
python scripts/pmc_traffic.py $tag $out/prof16_FETCH_SIZE $out/prof16_WRITE_SIZE _bf16
rm -rf $out/prof16_FETCH_SIZE $out/prof16_WRITE_SIZE
cp $out/${tag}_traffic.json $out/${tag}_traffic_bf16.json $R/profiles/
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_stats -- python3 $R/bench.py --no-extra --no-cpu-baseline > $out/${tag}_bench_under_rocprof.json 2> $out/prof_stats.err || { tail -5 $out/prof_stats.err; exit 1; }
cp $(find $out/prof_stats -name '*kernel_stats.csv' | head -1) $out/${tag}_bench_kernel_stats.csv
python $R/scripts/stats_family.py $out/${tag}_bench_kernel_stats.csv $out/${tag}_bench_under_rocprof.json > $out/${tag}_dominant_kernel.json
cd $R
cp $out/${tag}_dominant_kernel.json $R/profiles/           # the bench line below quotes its in-graph average (same sources: sha checked)
rm -rf $out/prof_stats
timeout -k 10 500 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err || { tail -5 $out/${tag}_bench.err; exit 1; }
# 4. matrix-pipe utilisation of the GEMM kernels (own PMC pass): SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs)
#    against SQ_BUSY_CYCLES (summed over the 32 shader engines) -> gpurun_out/<tag>_pmc_mfma_busy.json
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/prof_mfma -- python3 $R/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-extra --no-roofline > $out/prof_mfma.log 2>&1 || { tail -5 $out/prof_mfma.log; exit 1; }
cd $R
python scripts/pmc_mfma_busy.py $out/prof_mfma > $out/${tag}_pmc_mfma_busy.json
rm -rf $out/prof_mfma
# 5. the bf16 configuration (BASELINE configs[2]): bench line + kernel stats of `bench.py --dtype bf16`
timeout -k 10 300 python bench.py --dtype bf16 --no-extra --no-cpu-baseline > $out/${tag}_bench_bf16.json 2> $out/${tag}_bench_bf16.err || { tail -5 $out/${tag}_bench_bf16.err; exit 1; }
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bf16 -- python3 $R/bench.py --dtype bf16 --no-extra --no-cpu-baseline > $out/${tag}_bench_bf16_under_rocprof.json 2> $out/prof_bf16.err || { tail -5 $out/prof_bf16.err; exit 1; }
cp $(find $out/prof_bf16 -name '*kernel_stats.csv' | head -1) $out/${tag}_bf16_mode_kernel_stats.csv
rm -rf $out/prof_bf16
cd $R
# 6. the C5 tile shape (BASELINE configs[4]: NB 40, L 512), fp32 and bf16
timeout -k 10 300 python bench.py --nb 40 --seq-len 512 --no-cpu-baseline > $out/${tag}_bench_c5_f32.json 2> /dev/null
timeout -k 10 300 python bench.py --nb 40 --seq-len 512 --dtype bf16 --no-cpu-baseline > $out/${tag}_bench_c5_bf16.json 2> /dev/null
# 7. densenet18, the reference's DEFAULT backbone: bench line, kernel stats and the timeline of one captured step
timeout -k 10 300 python bench.py --backbone densenet18 --no-extra --no-cpu-baseline > $out/${tag}_bench_densenet18.json 2> $out/${tag}_bench_densenet18.err || { tail -5 $out/${tag}_bench_densenet18.err; exit 1; }
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_dn -- python3 $R/bench.py --backbone densenet18 --no-extra --no-cpu-baseline --no-roofline > $out/${tag}_bench_densenet18_under_rocprof.json 2> $out/prof_dn.err || { tail -5 $out/prof_dn.err; exit 1; }
cp $(find $out/prof_dn -name '*kernel_stats.csv' | head -1) $out/${tag}_densenet_kernel_stats.csv
rm -rf $out/prof_dn
cd $R
bash scripts/trace_round.sh ${tag}_densenet --backbone densenet18 > /dev/null
# 8. conv arithmetic 'f32x3p' (opt-in): bench line, kernel stats of the same command, per-layer timings against the fp32 kernels
timeout -k 10 300 python bench.py --dtype f32x3p --no-extra --no-cpu-baseline > $out/${tag}_bench_f32x3p.json 2> $out/${tag}_bench_f32x3p.err || { tail -5 $out/${tag}_bench_f32x3p.err; exit 1; }
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_x3p -- python3 $R/bench.py --dtype f32x3p --no-extra --no-cpu-baseline --no-roofline > $out/${tag}_bench_f32x3p_under_rocprof.json 2> $out/prof_x3p.err || { tail -5 $out/prof_x3p.err; exit 1; }
cp $(find $out/prof_x3p -name '*kernel_stats.csv' | head -1) $out/${tag}_f32x3p_kernel_stats.csv
rm -rf $out/prof_x3p
cd $R
{ python scripts/bench_x3p.py; python scripts/bench_x3p_s2.py; } 2>&1 | grep -v amdgpu.ids > $out/${tag}_x3p_micro.txt
