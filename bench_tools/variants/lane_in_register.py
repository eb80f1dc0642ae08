COMPILE = ["ntt_kernels.hip", "behz_kernels.hip"]  # the edited text lives in a header both include
DESCRIPTION = ("the steps' lane index carried in a vector register (an opaque copy, round 4's first form) instead of re-derived from "
               "the wave's base and the lane's position in its wave")
EDITS = [("ntt_rows.hpp", "    return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(lane))) + within;", "    (void)within;\n    return opaque32(lane);")]
