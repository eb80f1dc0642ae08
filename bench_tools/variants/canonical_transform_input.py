COMPILE = ["ntt_kernels.hip", "behz_kernels.hip"]  # the edited text lives in a header both include
DESCRIPTION = ("the fused inverse loads (tensor product, key MAC) reduce their words to [0, p) before the transform (rounds 2-4) "
               "instead of handing it the bounded reduction's remainder in [0, 5p)")
EDITS = [("ntt_rows.hpp", "constexpr bool kLazyTransformInput = is_split(MODE) || is_fold(MODE);",
          "constexpr bool kLazyTransformInput = false;")]
