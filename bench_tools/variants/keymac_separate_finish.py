DESCRIPTION = "the key switch ends in key_switch_finish_kernel (round 3) instead of in the key-MAC transform's store"
EDITS = [("ntt_kernels.hip", "    const bool tiled = ks.log_degree >= 12 && ks.log_degree <= 14;  // the degrees with a fused key-MAC transform",
          "    const bool tiled = false;")]
