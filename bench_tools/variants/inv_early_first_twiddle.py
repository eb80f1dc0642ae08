COMPILE = ["ntt_kernels.hip", "behz_kernels.hip"]  # the edited text lives in a header both include
DESCRIPTION = ("the first twiddle of every inverse pass on the limb-wise butterflies requested BEFORE the exchange that feeds "
               "the pass (with late lane addresses: 12 B of scratch in the plain-slab kernel instead of 0)")
EDITS = [("ntt_rows.hpp", "template <int MODE>\nconstexpr bool kInverseFirstTwiddleEarly = false;",
          "template <int MODE>\nconstexpr bool kInverseFirstTwiddleEarly = MODE == kModeSplit;")]
