COMPILE = ["ntt_kernels.hip"]
DESCRIPTION = "N = 32768: the ct x ct / key-switching pipelines unfused over the plain interleaved transforms (rounds 3-4)"
EDITS = [("ntt_kernels.hip", "constexpr uint32_t kMaxFusedLoadLogDegree = 15;", "constexpr uint32_t kMaxFusedLoadLogDegree = 14;")]
