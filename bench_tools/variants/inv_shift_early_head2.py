DESCRIPTION = ("inv_shift_early + two twiddles of the first (gather-heavy) pass requested before the rows are loaded and "
               "two kept in flight through that pass (12 B of scratch)")
EDITS = [("ntt_kernels.hip", "constexpr bool kShiftFactors = !INVERSE && LOGN == 12;",
          "constexpr bool kShiftFactors = (!INVERSE && LOGN == 12) || (INVERSE && LOGN == 13);"),
         ("ntt_kernels.hip", "template <int MODE>\nconstexpr bool kInverseFirstTwiddleEarly = false;",
          "template <int MODE>\nconstexpr bool kInverseFirstTwiddleEarly = MODE == kModeSplitShift;"),
         ("ntt_kernels.hip", "constexpr int kInverseHeadTwiddles = 1;",
          "constexpr int kInverseHeadTwiddles = (MODE == kModeSplitShift && SOURCE == kInverseFromSlab) ? 2 : 1;")]
