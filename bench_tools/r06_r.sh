cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r06r; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
python bench_tools/ab_variants.py run --what degrees --rounds 3 > $O/degrees.txt 2>&1; cat $O/degrees.txt
for lib in "" $PWD/swift-homomorphic-encryption_amd/lib/variants/libhe_amd_d_below_2_b33.so; do
  HEAMD_LIBRARY=$lib python bench_tools/param_sets_bench.py 2>/dev/null | tr '|' '\n' | grep "16384\|32768"
done
