DESCRIPTION = ("gathered twiddles of the limb-wise butterflies through flat addresses (scalar table base + 32-bit lane offset) "
               "instead of buffer descriptors")
COMPILE = ["ntt_kernels.hip"]
EDITS = [("ntt_common.hpp",
          "        const Dwordx4 pair = __builtin_amdgcn_raw_buffer_load_b128(tw.pair_resource, lane_index << 4, fixed_index << 4, 0);\n"
          "        const Dwordx2 factors = __builtin_amdgcn_raw_buffer_load_b64(tw.factor_resource, lane_index << 3, fixed_index << 3, 0);\n"
          "        t.w = pack64(pair.x, pair.y);\n"
          "        t.second = pack64(pair.z, pair.w);\n"
          "        t.factors = pack64(factors.x, factors.y);\n"
          "    } else if constexpr (is_fold(MODE) || MODE == kModeFoldLazy) {",
          "        const U64x2 pair = tw.pairs[size_t(lane_index + fixed_index)];\n"
          "        t.w = pair.x;\n"
          "        t.second = pair.y;\n"
          "        t.factors = tw.factors[size_t(lane_index + fixed_index)];\n"
          "    } else if constexpr (is_fold(MODE) || MODE == kModeFoldLazy) {")]
