COMPILE = ["behz_kernels.hip"]
DESCRIPTION = "the row-fused ct x ct kernels from 29 ciphertext pairs on (more than 256 (item, row) workgroups), as until round 6 (production: more than 1152 = 129 pairs at L = 4)"
EDITS = [("behz_kernels.hip", "constexpr size_t kBehzRowsFusedAbove = 1152;", "constexpr size_t kBehzRowsFusedAbove = 256;")]
