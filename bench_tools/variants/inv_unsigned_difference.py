COMPILE = ["ntt_kernels.hip"]
DESCRIPTION = ("every limb-wise inverse transform multiplies x + bound - y as an unsigned word (rounds 2-4: kModeSplit) instead of the "
               "signed difference x - y (kModeSplitSigned) where that is the committed form")
EDITS = [
    ("ntt_kernels.hip", "constexpr bool kSignedInverse = !(LOGN == 13 && (SOURCE == kInverseFromSlab || SOURCE == kInverseFromSlabScaled));",
     "constexpr bool kSignedInverse = false;"),
    ("ntt_kernels.hip", "constexpr int kInterleavedInverseSplit = kModeSplitSigned;", "constexpr int kInterleavedInverseSplit = kModeSplit;"),
]
