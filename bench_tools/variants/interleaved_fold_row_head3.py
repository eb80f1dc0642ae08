COMPILE = ["ntt_kernels.hip"]
DESCRIPTION = "... the sub-rows' inverse transform opens with three twiddles of its first pass in flight (production: one)"
EDITS = [("ntt_kernels.hip", "    TwiddleWords head[1];\n    inverse_row_head<kSubLogN, kSubLogE, MODE, false, 1>(head, tail, tid);\n    // (two sub-rows on the shift-folded products",
          "    constexpr int ROW_HEAD = MODE == kModeFoldLazy ? 3 : 1;\n    TwiddleWords head[ROW_HEAD];\n    inverse_row_head<kSubLogN, kSubLogE, MODE, false, ROW_HEAD>(head, tail, tid);\n    // (two sub-rows on the shift-folded products"),
         ("ntt_kernels.hip", "LOGS + INPUT_STAGES, LOGD, false, 1, LATE>(v, tid, tail, mod, lds, head);", "LOGS + INPUT_STAGES, LOGD, false, ROW_HEAD, LATE>(v, tid, tail, mod, lds, head);")]
