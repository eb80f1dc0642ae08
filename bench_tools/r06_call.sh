cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r06g; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for i in 1 2 3; do python bench_tools/path_bench.py --only=config3_ct_mul > $O/c3_$i.json 2>/dev/null; python - $O/c3_$i.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))["config3_ct_mul"]
print("ct_mul %.0f relin %.0f both %.0f" % (d["ct_mul_per_s"], d["relinearize_per_s"], d["ct_mul_relinearize_per_s"]), d["spread_ms"]["relinearize"])
PY
done
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-600 $O/bench.json
