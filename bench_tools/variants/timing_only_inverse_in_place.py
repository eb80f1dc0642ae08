COMPILE = ["ntt_kernels.hip"]
DESCRIPTION = "TIMING ONLY (the out-of-place callers read the wrong rows): the plain inverse kernels without the choice of a separate source slab -- what that choice costs the headline inverse transform"
EDITS = [("ntt_kernels.hip", "            if constexpr (SOURCE == kInverseFromSlab) load_base = tensor_source != nullptr ? tensor_source : slab;\n", "")]
