DESCRIPTION = ("round-3 key MAC: a workgroup pairs the same key column of two consecutive polynomials (the other column in a "
               "sibling workgroup of the same XCD), separate key_switch_finish_kernel")
EDITS = [("ntt_kernels.hip", "constexpr bool kKeyMacColumnPairs = true;", "constexpr bool kKeyMacColumnPairs = false;")]
