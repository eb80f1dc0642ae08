#!/bin/bash
# One GPU call of round 3: bash bench_tools/r03_call.sh TAG STEP [STEP ...]; every step writes under gpurun_out/TAG/.
cd "$GRAFT_REPO_ROOT"
T=$1; shift
O=gpurun_out/$T
mkdir -p $O
export TMPDIR=/tmp
for step in "$@"; do
  case $step in
    tests) timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log ;;
    ab_ntt) timeout 900 python bench_tools/ab_variants.py run --what degrees --rounds 2 > $O/ab_degrees.txt 2>&1; cat $O/ab_degrees.txt ;;
    ab_c3) timeout 900 python bench_tools/ab_variants.py run --what c3 --rounds 2 > $O/ab_c3.txt 2>&1; cat $O/ab_c3.txt ;;
    variant_parity) for lib in swift-homomorphic-encryption_amd/lib/variants/libhe_amd_*.so; do echo "== $lib"; HEAMD_LIBRARY=$PWD/$lib timeout 600 python -m pytest tests/test_gpu_ntt.py -m gpu -q -k "${PARITY_K:-4096-bits5 or 8192-bits6 or variants_agree or row_pairs or full_size}" 2>&1 | tail -4; done > $O/variant_parity.txt 2>&1; cat $O/variant_parity.txt ;;
    c4) timeout 600 python bench.py --workload c4 --steps 10 --warmup 2 > $O/bench_c4.json 2> $O/bench_c4.err; python -c "import json; r=json.load(open('$O/bench_c4.json')); print('c4', r['value'], r['roofline']['frac'], r['roofline']['avg_launch_ms'])" ;;
    elementwise) timeout 600 python bench_tools/elementwise_bench.py > $O/elementwise.txt 2>&1; tail -12 $O/elementwise.txt ;;
    bench) timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || tail -5 $O/bench.err; cut -c1-600 $O/bench.json ;;
    *) echo "unknown step $step" ;;
  esac
done
