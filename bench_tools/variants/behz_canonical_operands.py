DESCRIPTION = "row-fused BEHZ kernel: the transformed operands made canonical before the tensor product (one more conditional subtract per word)"
EDITS = [("behz_kernels.hip", "constexpr bool kBehzLazyOperands = true;", "constexpr bool kBehzLazyOperands = false;")]
