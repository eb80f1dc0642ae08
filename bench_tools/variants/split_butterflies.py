DESCRIPTION = "every transform of moduli 2^b - d on the limb-wise Shoup butterflies (8 multiply-adds, tabulated factors; the inverse in its signed form) instead of the fold-at-2^(b+2) ones -- the tree before r05ad"
EDITS = [
    ("ntt_kernels.hip", "constexpr bool kFoldLazyForward = (LOGN == 12 && LOGT == 9) || (LOGN == 13 && LOGT == 10);", "constexpr bool kFoldLazyForward = false;"),
    ("ntt_kernels.hip", "constexpr bool kFoldLazyInverse = (LOGN == 12 && LOGT == 9) || (LOGN == 13 && LOGT == 10);", "constexpr bool kFoldLazyInverse = false;"),
    ("behz_kernels.hip", "constexpr bool kBehzFoldLazy = true;", "constexpr bool kBehzFoldLazy = false;"),
]
