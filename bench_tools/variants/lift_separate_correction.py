DESCRIPTION = ("the lift's mTilde row as full exact sums and its correction term r (Q mod Bsk_j) as a Shoup product next to the "
               "sum's reduction (rounds 1-4) instead of low-word multiply-adds and one more product of the same exact sum")
EDITS = [
    ("rns_kernels.hip", "            if (sizeof(W) == 8 && last.p == kMTildeValue) {", "            if (false && sizeof(W) == 8 && last.p == kMTildeValue) {"),
    ("rns_kernels.hip", "                if constexpr (BOUNDED && sizeof(W) == 8) {\n                    // the r term is one more product",
     "                if constexpr (false && BOUNDED && sizeof(W) == 8) {\n                    // the r term is one more product"),
]
