#!/bin/bash
# inverse NTT A/B: production and the named variants, degrees 4096 / 8192 / 16384 (bench_tools/ab_variants.py run) and
# the ct x ct + relinearize pipeline.   bash bench_tools/inv_ab.sh variant [variant ...]
python bench_tools/ab_variants.py run --what degrees --rounds ${ROUNDS:-3} "$@"
python bench_tools/ab_variants.py run --what c3 --rounds ${ROUNDS:-3} "$@"
