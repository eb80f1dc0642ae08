#!/bin/bash
# Round-6 closing pass (run ONCE per round): parity, the driver's bench command (+ rocprofv3 kernel stats of the same command), HBM counters
# of the four workloads (-> profiles/r06_pmc_traffic.json, what bench.py reports as `traffic`), the other workload lines,
# next-row and UInt32 benches.   bash bench_tools/r06_final.sh TAG
cd "$GRAFT_REPO_ROOT"
T=${1:-r06z}
mkdir -p gpurun_out/$T
export TMPDIR=/tmp
O=gpurun_out/$T
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
pmc() {  # name target
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $c | cut -d' ' -f1)
    mkdir -p $O/pmc_$1
    timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$1/$n -- python $2 > $O/pmc_$1/$n.log 2>&1 || echo "pmc $1 $c failed"
  done
  python bench_tools/pmc_traffic.py $O/pmc_$1 > /dev/null
}
pmc c2 "bench_tools/ntt_profile_target.py 0"
pmc c3 bench_tools/c3_profile_target.py
pmc c4 bench_tools/c4_profile_target.py
pmc c5 bench_tools/c5_profile_target.py
(for d in c2 c3 c4 c5; do echo "== $d"; python bench_tools/pmc_traffic.py $O/pmc_$d | grep -v "at::\|rocclr"; done) > $O/pmc_traffic_per_kernel.txt
python bench_tools/traffic_json.py $O/pmc_c2 $O/pmc_c3 $O/pmc_c4 $O/pmc_c5 $O/r06_pmc_traffic.json > /dev/null
cp $O/r06_pmc_traffic.json profiles/r06_pmc_traffic.json
# the vector-ALU instruction stream of ct x ct + relinearize (bench.py valu_roofline replays it) and this box's instruction rates
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_WAVES --output-format csv -d $O/valu/pass1 -- python bench_tools/c3_profile_target.py > $O/valu_pass1.log 2>&1 || echo "valu pass failed"
(for k in lift_kernel behz_rows_fused floor_kernel ntt_forward_tiled ntt_inverse_tiled; do python bench_tools/pmc_summary.py $O/valu $k; done) > $O/c3_valu_counters.txt 2>&1
python bench_tools/valu_json.py $O/c3_valu_counters.txt $O/r06_c3_valu.json && cp $O/r06_c3_valu.json profiles/r06_c3_valu.json && cp $O/c3_valu_counters.txt profiles/r06z_c3_valu_counters.txt
sed -i 's#"source": "[^"]*"#"source": "profiles/r06z_c3_valu_counters.txt (rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 over bench_tools/c3_profile_target.py, closing pass)"#' profiles/r06_c3_valu.json
rm -rf $O/valu
./bench_tools/microbench > $O/microbench.txt 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || tail -5 $O/bench.err
cut -c1-400 $O/bench.json
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- python bench.py --gpus 1 --steps 20 --warmup 5 --skip-other-configs --no-cpu-baseline > $O/bench_ntt_only.json 2> $O/bench_stats.err
f=$(find $O/bench_stats -name "*kernel_stats.csv" | head -1); cp "$f" $O/bench_ntt_only_kernel_stats.csv; python bench_tools/kernel_stats_summary.py $O/bench_ntt_only_kernel_stats.csv | head -6
for w in c3 c4 c5; do
  timeout 600 python bench.py --workload $w --steps 10 --warmup 2 > $O/bench_$w.json 2> $O/bench_$w.err || tail -5 $O/bench_$w.err
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3_stats -- python bench_tools/c3_profile_target.py > $O/c3_stats.log 2>&1
f=$(find $O/c3_stats -name "*kernel_stats.csv" | head -1); cp "$f" $O/c3_kernel_stats.csv
timeout 600 python bench.py --workload c5 --device-group 2 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c5_group2.json 2> $O/bench_c5_group2.err || tail -5 $O/bench_c5_group2.err
timeout 600 python bench_tools/whole_query_spread.py after_legs 30 > $O/whole_query_spread.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/pir_trace -- python bench_tools/pir_loop_profile_target.py > $O/pir_loop.txt 2>&1
f=$(find $O/pir_trace -name "*kernel_trace.csv" | head -1); python bench_tools/kernel_gaps.py "$f" inner_product_plain_rows_kernel 40 > $O/pir_loop_timeline.txt 2>&1; rm -rf $O/pir_trace
timeout 600 python bench_tools/next_rows_bench.py > $O/next_rows.txt 2>&1
timeout 600 python bench_tools/word32_scheme_bench.py > $O/word32_scheme.json 2>&1
timeout 600 python bench_tools/ntt_variants.py > $O/ntt_variants.txt 2>&1
timeout 600 python bench_tools/expand_batch_profile_target.py > $O/expand_batch.txt 2>&1
timeout 600 python bench_tools/wire_format_bench.py > $O/wire_format.json 2>&1
timeout 300 python bench_tools/host_seam_probe.py > $O/host_seam.txt 2>&1
timeout 600 python bench_tools/param_sets_bench.py > $O/param_sets.txt 2>&1
# register / scratch figures of the built library's headline, key-MAC and interleaved kernels, and the static instruction
# mix of the headline pair (the build's own code objects; no GPU involved)
python bench_tools/kernel_metadata.py swift-homomorphic-encryption_amd/csrc/build/ntt_kernels.o --filter "<13, 10, 3" > $O/isa_stats.txt 2>&1
python bench_tools/kernel_metadata.py swift-homomorphic-encryption_amd/csrc/build/ntt_kernels.o --filter "<13, 10, 7" >> $O/isa_stats.txt 2>&1
python bench_tools/kernel_metadata.py swift-homomorphic-encryption_amd/csrc/build/ntt_kernels.o --filter "<13, 10, 4" >> $O/isa_stats.txt 2>&1
python bench_tools/kernel_metadata.py swift-homomorphic-encryption_amd/csrc/build/ntt_kernels.o --filter "<12, 9, " >> $O/isa_stats.txt 2>&1
python bench_tools/kernel_metadata.py swift-homomorphic-encryption_amd/csrc/build/ntt_kernels.o --filter "interleaved" >> $O/isa_stats.txt 2>&1
python bench_tools/kernel_metadata.py swift-homomorphic-encryption_amd/csrc/build/behz_kernels.o --filter "behz" >> $O/isa_stats.txt 2>&1
python bench_tools/kernel_metadata.py --spills-only >> $O/isa_stats.txt 2>&1
rm -rf $O/bench_stats $O/c3_stats $O/pmc_c2/*/ $O/pmc_c3/*/ $O/pmc_c4/*/ $O/pmc_c5/*/
