DESCRIPTION = ("rounds 2-3: every step's lane addresses are derived from the lane index wherever the compiler likes -- at the "
               "top of the kernel, carried through the passes in scratch where the register file is full")
EDITS = [("ntt_kernels.hip", "constexpr bool kLateLaneAddresses = true;", "constexpr bool kLateLaneAddresses = false;")]
