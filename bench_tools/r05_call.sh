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
  timeout ${PYTEST_TIMEOUT:-1500} python -m pytest tests -m gpu -q -s -x "${sel[@]}" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
  grep -E "compared with the oracle|passed|failed|rc=" $O/pytest.log | tail -12
fi
for leg in ${LEGS:-c3ab c3table}; do
  case $leg in
    c3ab) timeout 900 python bench_tools/ab_variants.py run --what c3 --rounds ${ROUNDS:-3} ${C3_VARIANTS:-} > $O/ab_c3.txt 2>&1; cat $O/ab_c3.txt ;;
    nttab) timeout 900 python bench_tools/ab_variants.py run --what ${NTT_WHAT:-ntt} --rounds ${ROUNDS:-3} ${NTT_VARIANTS:-} > $O/ab_ntt.txt 2>&1; cat $O/ab_ntt.txt ;;
    c3table) bash bench_tools/c3_before_after.sh $T ${VARIANT:-behz_unfused_rows} ;;
    bench) timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json ;;
    c3bench) timeout 600 python bench.py --workload c3 > $O/bench_c3.json 2> $O/bench_c3.err; cat $O/bench_c3.json ;;
    *) echo "unknown leg $leg" ;;
  esac
done
