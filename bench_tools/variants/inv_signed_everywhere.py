COMPILE = ["ntt_kernels.hip"]
DESCRIPTION = ("the plain-slab inverse transform at N = 8192 in the signed form too (kModeSplitSigned, lane index re-derived), as the "
               "other limb-wise inverse kernels")
EDITS = [
    ("ntt_kernels.hip", "constexpr bool kSignedInverse = !(LOGN == 13 && (SOURCE == kInverseFromSlab || SOURCE == kInverseFromSlabScaled));",
     "constexpr bool kSignedInverse = true;"),
]
