#!/bin/bash
# usage: tools/evidence.sh <tag> <part a|b|c>   (on the GPU box, from the repo root; one part per gpurun call: each fits the call limit)
# Collects what profiles/<tag>/ holds.
#   a: the bench line, the one-rank RCCL line, rocprofv3 kernel stats of the bench command, the four PMC passes, stage times
#   b: config 5 through the CLI, source text, per-kernel profiles of the deep-LCP shapes, per-round traces, two contexts, long copies / defects
#   c: small blocks (batch entry points), inverse chain heads, validate timing, post stage, micro-benchmarks
set -o pipefail
tag=$1; part=${2:-a}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
if [ $part = a ]; then
timeout -k 10 400 python3 bench.py --steps 20 --warmup 2 > $out/bench_line.json 2> $out/bench.err || exit 1
echo "bench done"
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 2 --no-cpu-baseline > $out/bench_line_torchrun_world1.json 2> $out/bench_w1.err || exit 1
timeout -k 10 300 python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-shapes > $out/bench_line_same_box_again.json 2>> $out/bench.err || exit 1
echo "world-1 done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-shapes > $out/stats.log 2>&1 || exit 1
cp $(ls $out/stats/*/*_kernel_stats.csv | tail -1) $out/kernel_stats.csv
echo "stats done"
bash tools/pmc.sh $tag/pmc fetch FETCH_SIZE -- 256 random 3 || exit 1
bash tools/pmc.sh $tag/pmc write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- 256 random 3 || exit 1
bash tools/pmc.sh $tag/pmc sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS -- 256 random 3 || exit 1
bash tools/pmc.sh $tag/pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU -- 256 random 3 || exit 1
python3 tools/pmc_summary.py $out/pmc > $out/pmc_fetch_write.txt 2>&1
python3 tools/pmc_traffic.py $out/pmc profiles/$tag/pmc_fetch_write.txt $out/pmc_traffic.json > $out/pmc_traffic.log 2>&1
echo "pmc done"
timeout -k 10 200 python3 tools/pcie_inclusive.py 256 2>/dev/null | tail -1 > $out/pcie_inclusive.json; cat $out/pcie_inclusive.json
for sh in random dna text a ab motif prose motif_defects random_copy; do
  timeout -k 10 120 python3 tools/stage_times.py 256 $sh 3 2>/dev/null | tail -1 | sed "s/^/$sh /" >> $out/stage_times.txt || exit 1
done
timeout -k 10 120 python3 tools/stage_times.py 256 random 3 inv 2>/dev/null | tail -1 | sed "s/^/inverse-random /" >> $out/stage_times.txt
for mib in 4 16 64 128; do
  timeout -k 10 120 python3 tools/stage_times.py $mib random 3 2>/dev/null | tail -1 | sed "s/^/forward-random-${mib}MiB /" >> $out/stage_times.txt
  timeout -k 10 120 python3 tools/stage_times.py $mib random 3 inv 2>/dev/null | tail -1 | sed "s/^/inverse-random-${mib}MiB /" >> $out/stage_times.txt
done
cat $out/stage_times.txt
fi
if [ $part = b ]; then
timeout -k 10 600 python3 tools/config5.py 256 2>/dev/null | tail -1 > $out/config5.json && cat $out/config5.json
timeout -k 10 300 python3 tools/real_text.py 256 2>/dev/null | tail -1 > $out/real_text.json && cat $out/real_text.json
bash tools/prof_kernels.sh real > /dev/null 2>&1 && cp gpurun_out/prof_real.txt $out/kernels_real_text.txt
bash tools/prof_kernels.sh prose > /dev/null 2>&1 && cp gpurun_out/prof_prose.txt $out/kernels_prose.txt
bash tools/prof_kernels.sh text > /dev/null 2>&1 && cp gpurun_out/prof_text.txt $out/kernels_text.txt
bash tools/prof_kernels.sh motif_defects > /dev/null 2>&1 && cp gpurun_out/prof_motif_defects.txt $out/kernels_motif_defects.txt
echo "kernel profiles done"
if [ -f dark-archon_amd/libarchon_hip_exp.so ]; then
  ARCHON_HIP_LIB=$PWD/dark-archon_amd/libarchon_hip_exp.so ARCHON_TRACE_ROUNDS=1 timeout -k 10 200 python3 tools/stage_times.py 256 prose 2 2>&1 | grep "^round\|general_stage" | tail -24 > $out/rounds_prose.txt
  ARCHON_HIP_LIB=$PWD/dark-archon_amd/libarchon_hip_exp.so ARCHON_TRACE_ROUNDS=1 timeout -k 10 300 python3 tools/real_text.py 256 2>&1 | grep "^round\|general_stage" | tail -24 > $out/rounds_real_text.txt
fi
timeout -k 10 300 python3 tools/two_ctx.py 256 6 > $out/two_contexts.txt 2>/dev/null; cat $out/two_contexts.txt
timeout -k 10 200 python3 tools/dup_region.py 256 32 2>/dev/null | tail -1 > $out/dup_region_256_32.json
timeout -k 10 300 python3 tools/defect_motif.py 80 3 2>/dev/null | tail -1 > $out/defect_motif_80_3.json
timeout -k 10 120 python3 tools/radix_dir_bench.py 2>/dev/null | tail -1 > $out/radix_dir_bench.json && cat $out/radix_dir_bench.json
fi
if [ $part = c ]; then
timeout -k 10 400 python3 tools/small_blocks.py 1 4 16 64 2>/dev/null > $out/small_blocks.txt; cat $out/small_blocks.txt
ARCHON_SMALL_BLOCK=0 timeout -k 10 200 python3 tools/small_blocks.py 4 2>/dev/null | sed 's/^/streaming-stage-forced /' >> $out/small_blocks.txt
timeout -k 10 300 python3 tools/inv_sbits_sweep.py 2>/dev/null > $out/inv_sbits_sweep.txt; cat $out/inv_sbits_sweep.txt
timeout -k 10 300 python3 tools/validate_timing.py 256 2>/dev/null | tail -1 > $out/validate_timing.json; cat $out/validate_timing.json
timeout -k 10 300 python3 tools/post_bench.py 256 > $out/post_stage.txt 2>/dev/null
hipcc -O3 --offload-arch=gfx950 -o /tmp/gather_chain tools/micro/gather_chain.hip 2>/dev/null && timeout -k 10 300 /tmp/gather_chain > $out/micro_gather_chain.txt 2>&1
timeout -k 10 200 python3 tools/inv_exp.py 2>/dev/null | tail -5 > $out/inverse_walk_parts.txt
bash tools/inv_kernels.sh 8 > $out/inverse_kernels.txt 2>&1
timeout -k 10 300 python3 tools/post_decode_bench.py 256 > $out/post_decode.txt 2>/dev/null
for m in lds_rates pass_model scatter_pass lf_build; do
  hipcc -O3 --offload-arch=gfx950 -o /tmp/$m tools/micro/$m.hip 2>/dev/null && timeout -k 10 120 /tmp/$m > $out/micro_$m.txt 2>&1
done
fi
echo "evidence part $part complete"
