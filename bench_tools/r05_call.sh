#!/bin/bash
# Round-5 GPU call: parity first (all GPU tests, the exhaustive full-size ones included), then the legs named in LEGS:
#   c3ab     production against every variant library under lib/variants/ on ct x ct + relinearize (interleaved rounds)
#   c3table  rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE per kernel, variant $VARIANT (before) against production (after)
#   nttab    the same A/B on the headline transforms            ntttable  PMC passes of the headline kernels
#   bench    the driver's command                               host      the PCIe-inclusive host seam
# bash bench_tools/r05_call.sh TAG        (LEGS="c3ab c3table" VARIANT=behz_unfused_rows PYTEST="-k mul" ...)
cd "$GRAFT_REPO_ROOT"
T=${1:-r05a}
O=gpurun_out/$T
mkdir -p $O
export TMPDIR=/tmp
if [ "${PYTEST:-all}" != "none" ]; then
  sel=(); [ "${PYTEST:-all}" != "all" ] && sel=(-k "$PYTEST")
  timeout ${PYTEST_TIMEOUT:-1500} python -m pytest tests -m gpu -q -s ${PYTEST_FLAGS:--x} "${sel[@]}" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
  grep -E "compared with the oracle|passed|failed|rc=" $O/pytest.log | tail -12
fi
for leg in ${LEGS-c3ab c3table}; do
  case $leg in
    c3ab) timeout 900 python bench_tools/ab_variants.py run --what c3 --rounds ${ROUNDS:-3} ${C3_VARIANTS:-} > $O/ab_c3.txt 2>&1; cat $O/ab_c3.txt ;;
    nttab) timeout 900 python bench_tools/ab_variants.py run --what ${NTT_WHAT:-ntt} --rounds ${ROUNDS:-3} ${NTT_VARIANTS:-} > $O/ab_ntt.txt 2>&1; cat $O/ab_ntt.txt ;;
    c3table) bash bench_tools/c3_before_after.sh $T ${VARIANT:-behz_unfused_rows} ;;
    bench) timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json ;;
    c3bench) timeout 600 python bench.py --workload c3 > $O/bench_c3.json 2> $O/bench_c3.err; cat $O/bench_c3.json ;;
    c3pmc)  # issue / stall / LDS / instruction-cache counters of the ct x ct pipeline's kernels (PMC_LIB: a variant library)
      i=0
      for g in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
               "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
               "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVES SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
               "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_THREAD_CYCLES_VALU"; do
        i=$((i+1))
        env ${PMC_LIB:+HEAMD_LIBRARY=$PWD/$PMC_LIB} timeout 300 rocprofv3 --pmc $g --output-format csv -d $O/pmc/pass$i -- python bench_tools/c3_profile_target.py > $O/pmc_pass$i.log 2>&1 || echo "pmc pass $i failed"
      done
      for needle in ${PMC_NEEDLES:-behz_rows_fused ntt_forward_tiled ntt_inverse_tiled}; do python bench_tools/pmc_summary.py $O/pmc $needle; done > $O/c3_pmc_summary.txt 2>&1
      rm -rf $O/pmc; cat $O/c3_pmc_summary.txt ;;
    host)  # the host-pointer seam per call, production and the host_seam_* variant libraries
      (echo "production"; timeout 300 python bench_tools/host_seam_probe.py
       for lib in swift-homomorphic-encryption_amd/lib/variants/libhe_amd_host_seam_*.so; do
         [ -e "$lib" ] && { echo "$(basename $lib .so | sed s/libhe_amd_//)"; HEAMD_LIBRARY=$PWD/$lib timeout 300 python bench_tools/host_seam_probe.py; }
       done) > $O/host_seam.txt 2>&1; cat $O/host_seam.txt ;;
    params)  # the reference's 60-bit parameter sets and N = 16384, production and the variant libraries named in PARAM_VARIANTS
      (echo "production: $(timeout 600 python bench_tools/param_sets_bench.py 2>&1 | tail -1)"
       for v in ${PARAM_VARIANTS:-}; do
         lib=swift-homomorphic-encryption_amd/lib/variants/libhe_amd_$v.so
         [ -e "$lib" ] && echo "$v: $(HEAMD_LIBRARY=$PWD/$lib timeout 600 python bench_tools/param_sets_bench.py 2>&1 | tail -1)"
       done) > $O/param_sets.txt; cat $O/param_sets.txt ;;
    smallmul)  # ct x ct on small batches, production and the variant libraries named in SMALL_VARIANTS
      (echo "== production"; timeout 300 python bench_tools/small_batch_mul_bench.py 2>&1 | grep batch
       for v in ${SMALL_VARIANTS:-}; do
         lib=swift-homomorphic-encryption_amd/lib/variants/libhe_amd_$v.so
         [ -e "$lib" ] && { echo "== $v"; HEAMD_LIBRARY=$PWD/$lib timeout 300 python bench_tools/small_batch_mul_bench.py 2>&1 | grep batch; }
       done) > $O/small_batch_mul.txt; cat $O/small_batch_mul.txt ;;
    *) echo "unknown leg $leg" ;;
  esac
done
