COMPILE = ["ntt_kernels.hip"]
DESCRIPTION = "... one in flight, the inverse's first one requested before the row is loaded"
EDITS = [("ntt_kernels.hip", 'constexpr bool kCrossEarlySplit = false;', 'constexpr bool kCrossEarlySplit = true;')]
