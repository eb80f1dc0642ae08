COMPILE = ["ntt_kernels.hip"]
DESCRIPTION = "the cross stages of the interleaved rows (N = 16384 / 32768) on the limb-wise products with two twiddles in flight instead of one (6 registers each: 16 / 40 B of scratch forward / inverse)"
EDITS = [("ntt_kernels.hip", 'constexpr int kCrossAheadSplit = 1;', 'constexpr int kCrossAheadSplit = 2;')]
