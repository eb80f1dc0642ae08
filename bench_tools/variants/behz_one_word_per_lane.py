DESCRIPTION = ("the bounded lift on 8-byte slabs with one coefficient per lane (rounds 1-4) instead of two (one 16-byte access "
               "per lane and row, scalar constants and addresses shared by two coefficients)")
EDITS = [("rns_kernels.hip", "constexpr bool kLiftPairs = true;", "constexpr bool kLiftPairs = false;")]
