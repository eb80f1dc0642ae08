COMPILE = ["ntt_kernels.hip"]
DESCRIPTION = "TIMING ONLY (wrong results): the gathered twiddles of the limb-wise butterflies (the interleaved rows of N = 16384 / 32768) made up from the lane index instead of loaded -- what their gathers cost"
EDITS = [("ntt_common.hpp",
          "    } else if constexpr (MODE == kModeSplit || MODE == kModeSplitSigned) {\n"
          "        const Dwordx4 pair = __builtin_amdgcn_raw_buffer_load_b128(tw.pair_resource, lane_index << 4, fixed_index << 4, 0);\n"
          "        const Dwordx2 factors = __builtin_amdgcn_raw_buffer_load_b64(tw.factor_resource, lane_index << 3, fixed_index << 3, 0);\n"
          "        t.w = pack64(pair.x, pair.y);\n"
          "        t.second = pack64(pair.z, pair.w);\n"
          "        t.factors = pack64(factors.x, factors.y);\n",
          "    } else if constexpr (MODE == kModeSplit || MODE == kModeSplitSigned) {\n"
          "        t.w = pack64(lane_index * 2654435761u + fixed_index, lane_index & 0x3fffffu);\n"
          "        t.second = pack64(lane_index * 40503u + fixed_index, (lane_index >> 3) & 0x3fffffu);\n"
          "        t.factors = pack64(lane_index * 7919u + fixed_index, (lane_index >> 2) & 0xfffffu);\n")]
