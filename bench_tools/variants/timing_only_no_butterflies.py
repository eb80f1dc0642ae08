COMPILE = ["ntt_kernels.hip"]
DESCRIPTION = ("TIMING ONLY (wrong results): the forward transform on the shift-folded products with every butterfly's product and "
               "sums replaced by two 64-bit adds of the twiddle's words (the rows, the LDS exchanges and the twiddle gathers "
               "stay) -- the transform's traffic without its multiplies, for the power model (bench_tools/power_probe.py)")
EDITS = [("ntt_common.hpp",
          "        const uint64_t r = uniform ? fold_mul<true, false>(y, w.w, w.second, fc) : fold_mul<false, false>(y, w.w, w.second, fc);\n"
          "        first = x + r;\n"
          "        second = x + half_bound - r;\n"
          "        return;\n",
          "        first = x + w.w;\n"
          "        second = y + w.second;\n"
          "        return;\n")]
