DESCRIPTION = ("inv_shift + the first twiddle of every inverse pass requested BEFORE the exchange that feeds the pass "
               "(no scratch with 4-register twiddles: 0 B against 8 B)")
EDITS = [("ntt_kernels.hip", "constexpr bool kShiftFactors = !INVERSE && LOGN == 12;",
          "constexpr bool kShiftFactors = (!INVERSE && LOGN == 12) || (INVERSE && LOGN == 13);"),
         ("ntt_kernels.hip", "template <int MODE>\nconstexpr bool kInverseFirstTwiddleEarly = false;",
          "template <int MODE>\nconstexpr bool kInverseFirstTwiddleEarly = MODE == kModeSplitShift;")]
