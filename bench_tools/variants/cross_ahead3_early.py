COMPILE = ["ntt_kernels.hip"]
DESCRIPTION = "... three in flight, the inverse's first three requested before the row is loaded"
EDITS = [("ntt_kernels.hip", 'constexpr int kCrossAheadSplit = 1;', 'constexpr int kCrossAheadSplit = 3;'), ("ntt_kernels.hip", 'constexpr bool kCrossEarlySplit = false;', 'constexpr bool kCrossEarlySplit = true;')]
