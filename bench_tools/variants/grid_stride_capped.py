COMPILE = ["poly_kernels.hip", "galois_kernels.hip", "rns_kernels.hip", "word32_kernels.hip"]
DESCRIPTION = "the element-wise kernels on 256 x 8 (rns_kernels: 256 x 16) workgroups walking their items with a grid-stride loop, as until round 6 (production: one workgroup per 256 items)"
EDITS = [("poly_kernels.hip", "constexpr size_t kGridCap = (size_t(1) << 31) - 1;", "constexpr size_t kGridCap = 256 * 8;"),
         ("galois_kernels.hip", "constexpr size_t kGridCap = (size_t(1) << 31) - 1;", "constexpr size_t kGridCap = 256 * 8;"),
         ("rns_kernels.hip", "constexpr size_t kGridCap = (size_t(1) << 31) - 1;", "constexpr size_t kGridCap = 256 * 16;"),
         ("word32_kernels.hip", "constexpr size_t kGridCap = (size_t(1) << 31) - 1;", "constexpr size_t kGridCap = 256 * 8;")]
