COMPILE = ["ntt_kernels.hip", "behz_kernels.hip"]  # the edited text lives in a header both include
DESCRIPTION = ("rounds 2-3: every step's lane addresses are derived from the lane index wherever the compiler likes -- at the "
               "top of the kernel, carried through the passes in scratch where the register file is full")
EDITS = [("ntt_rows.hpp", "constexpr bool kLateLaneAddresses = true;", "constexpr bool kLateLaneAddresses = false;")]
