DESCRIPTION = ("key MAC: a workgroup takes the two key COLUMNS of one polynomial (every spread word fetched once), one word at a "
               "time -- two sums of 7 registers; a 16-byte pair per lane would need four -- through scalar buffer descriptors")
_MAP_OLD = """    if constexpr (!FROM_SLAB) {
        // records (item, c) of one item read the same source rows: one replica set per (group of ROWS consecutive"""
_MAP_NEW = """    if constexpr (KEYMAC && ROWS == 2) {
        // one workgroup per (polynomial, band row): records (polynomial, 0) and (polynomial, 1); record_base counts items
        uint32_t group;
        locate(map, blockIdx.x, group, within);
        record = (map.record_base + group) * 2;
#pragma unroll
        for (int k = 0; k < ROWS; ++k) rows[k] = size_t(record + k) * map.record_rows + map.band_offset + within;
    } else if constexpr (!FROM_SLAB) {
        // records (item, c) of one item read the same source rows: one replica set per (group of ROWS consecutive"""
_LOAD_OLD = """        } else if constexpr (KEYMAC) {
            const size_t first_poly = record >> 1, c = record & 1;  // record = poly * 2 + c"""
_LOAD_NEW = """        } else if constexpr (KEYMAC && ROWS == 2) {
            const uint32_t L = source_spec.L, top_rows = source_spec.top_rows;
            const uint32_t r = map.band_offset + within;
            const uint32_t key_row = (r == L) ? top_rows - 1 : r;
            const size_t poly = record >> 1;
            const uint32_t lane_bytes = lane_part<LOGN, LOGE, 0, LOW>(tid) << 3;
            const uint32_t spread_step = (L + 1) << (LOGN + 3), key_step = (2 * top_rows) << (LOGN + 3),
                           column_step = top_rows << (LOGN + 3);
            const BufferResource spread_rows = make_uniform_resource(source_spec.first + ((poly * L * (L + 1) + r) << LOGN),
                                                                     (L - 1) * spread_step + (8u << LOGN));
            const BufferResource key_rows = make_uniform_resource(source_spec.second + (static_cast<size_t>(key_row) << LOGN),
                                                                  (L - 1) * key_step + column_step + (8u << LOGN));
            const bool bounded = kKeyMacBoundedReduce && mod.wide_shift != 0 && L <= 8 && uint64_t(L) * mod.p < (uint64_t(1) << 63);
            auto word = [](Dwordx2 w) { return pack64(w.x, w.y); };
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const uint32_t at = register_part<LOGN, LOGE, 0, LOW>(e) << 3;
                uint64_t x = word(__builtin_amdgcn_raw_buffer_load_b64(spread_rows, lane_bytes, at, 0));
                uint64_t k0 = word(__builtin_amdgcn_raw_buffer_load_b64(key_rows, lane_bytes, at, 0));
                uint64_t k1 = word(__builtin_amdgcn_raw_buffer_load_b64(key_rows, lane_bytes, at + column_step, 0));
                ProductSum acc0 = product_sum_zero(), acc1 = product_sum_zero();
                for (uint32_t j = 0; j < L; ++j) {
                    const uint32_t ahead = j + 1 < L ? j + 1 : j;
                    const uint64_t xn = word(__builtin_amdgcn_raw_buffer_load_b64(spread_rows, lane_bytes, ahead * spread_step + at, 0));
                    const uint64_t k0n = word(__builtin_amdgcn_raw_buffer_load_b64(key_rows, lane_bytes, ahead * key_step + at, 0));
                    const uint64_t k1n = word(__builtin_amdgcn_raw_buffer_load_b64(key_rows, lane_bytes, ahead * key_step + column_step + at, 0));
                    product_sum_add_pair<is_split(MODE)>(acc0, acc1, k0, k1, x);
                    x = xn;
                    k0 = k0n;
                    k1 = k1n;
                }
                if (bounded) {
                    v[0][e] = reduce_product_sum_bounded(acc0, mod);
                    v[1][e] = reduce_product_sum_bounded(acc1, mod);
                } else {
                    v[0][e] = reduce_product_sum(acc0, mod);
                    v[1][e] = reduce_product_sum(acc1, mod);
                }
            }
        } else if constexpr (KEYMAC) {
            const size_t first_poly = record >> 1, c = record & 1;  // record = poly * 2 + c"""
_PC_OLD = "                const size_t pc = record + 2 * k;  // polynomial * 2 + c (the rows are the same column of consecutive polynomials)"
_PC_NEW = "                const size_t pc = record + (ROWS == 2 ? k : 0);"
_LAUNCH_OLD = """    if (source == kInverseFromTensor || is_key_mac(source)) {
        // records are (item, c): groups of consecutive ITEMS share a workgroup"""
_LAUNCH_NEW = """    if (is_key_mac(source) && kKeyMacRows<LOGN, LOGT> == 2) {
        const size_t workgroups = rows / 2;
        return source == kInverseFromKeyMacFinish
                   ? launch_inverse_kernel<LOGN, LOGT, kInverseFromKeyMacFinish, 2>(mode, slab, ctx, map, workgroups, source_spec, stream)
                   : launch_inverse_kernel<LOGN, LOGT, kInverseFromKeyMac, 2>(mode, slab, ctx, map, workgroups, source_spec, stream);
    }
    if (source == kInverseFromTensor || is_key_mac(source)) {
        // records are (item, c): groups of consecutive ITEMS share a workgroup"""
EDITS = [("ntt_kernels.hip", _MAP_OLD, _MAP_NEW), ("ntt_kernels.hip", _LOAD_OLD, _LOAD_NEW), ("ntt_kernels.hip", _PC_OLD, _PC_NEW),
         ("ntt_kernels.hip", _LAUNCH_OLD, _LAUNCH_NEW)]
