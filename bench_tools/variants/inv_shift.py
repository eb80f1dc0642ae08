DESCRIPTION = "plain-slab inverse NTT at N = 8192 on the shifted-factor butterflies (a gathered twiddle is 4 registers instead of 6)"
EDITS = [("ntt_kernels.hip", "constexpr bool kShiftFactors = !INVERSE && LOGN == 12;",
          "constexpr bool kShiftFactors = (!INVERSE && LOGN == 12) || (INVERSE && LOGN == 13);")]
