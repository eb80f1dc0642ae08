DESCRIPTION = "key MAC pairs the two key columns of one polynomial, but the key switch still ends in key_switch_finish_kernel"
EDITS = [("ntt_kernels.hip", "    const bool paired = kKeyMacColumnPairs && (ks.log_degree == 12 || ks.log_degree == 13);  // kKeyMacRows == 2",
          "    const bool paired = false;")]
