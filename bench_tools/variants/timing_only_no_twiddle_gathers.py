DESCRIPTION = "TIMING ONLY (wrong results): the gathered twiddles of the shift-folded butterflies made up from the lane index instead of loaded -- what the gathers cost"
EDITS = [("ntt_common.hpp",
          "    } else if constexpr (is_fold(MODE) || MODE == kModeFoldLazy) {\n"
          "        const Dwordx4 pair = __builtin_amdgcn_raw_buffer_load_b128(tw.pair_resource, lane_index << 4, fixed_index << 4, 0);\n"
          "        t.w = pack64(pair.x, pair.y);\n"
          "        t.second = pack64(pair.z, pair.w);\n",
          "    } else if constexpr (is_fold(MODE) || MODE == kModeFoldLazy) {\n"
          "        t.w = pack64(lane_index * 2654435761u + fixed_index, lane_index & 0x3fffffu);\n"
          "        t.second = pack64(lane_index * 40503u + fixed_index, (lane_index >> 3) & 0x3fffffu);\n")]
COMPILE = ["ntt_kernels.hip", "behz_kernels.hip"]
