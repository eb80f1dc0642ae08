DESCRIPTION = ("the key switch's fused transforms (decomposition into the forward NTT, inner product with the key into the "
               "inverse NTT) as one launch over all key-switching moduli on the [0, 8p) butterflies (rounds 2-3)")
EDITS = [("ntt_kernels.hip", "    if constexpr (kFoldShape<LOGN, LOGT> && (SPREAD == kSourceSlab || SPREAD == kSourceSpread)) {",
          "    if constexpr (kFoldShape<LOGN, LOGT> && SPREAD == kSourceSlab) {"),
         ("ntt_kernels.hip", "    if constexpr (kFoldShape<LOGN, LOGT>) {\n        if (mode == kModeApprox && ctx.forward_split_pairs != nullptr) {\n            const int fold = fold_mode(ctx, map.mod_base, map.band_rows);\n            if (fold == kModeFoldMinus) kernel = ntt_inverse_tiled",
          "    if constexpr (kFoldShape<LOGN, LOGT> && !is_key_mac(SOURCE)) {\n        if (mode == kModeApprox && ctx.forward_split_pairs != nullptr) {\n            const int fold = fold_mode(ctx, map.mod_base, map.band_rows);\n            if (fold == kModeFoldMinus) kernel = ntt_inverse_tiled"),
         ("ntt_kernels.hip", "    const int count = (ks_ctx.log_degree == 12 || ks_ctx.log_degree == 13) && rows > kOneGeneration ? band_runs(ks_ctx, period, runs) : 0;",
          "    const int count = 0;"),
         ("ntt_kernels.hip", "    const int run_count = fold_shape && records * count > kOneGeneration ? band_runs(ks, L + 1, runs) : 0;",
          "    const int run_count = 0 * int(fold_shape);")]
