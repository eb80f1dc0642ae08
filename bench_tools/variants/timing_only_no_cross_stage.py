COMPILE = ["ntt_kernels.hip"]
DESCRIPTION = "TIMING ONLY (wrong results): the interleaved rows of N = 16384 / 32768 without their cross stages (the stages that pair words of different sub-rows: 8 gathered twiddles per lane and stage) -- what the extra stages cost"
EDITS = [("ntt_kernels.hip", "    forward_cross_stages<LOGS, MODE>(v, tid, tw, p);\n    canonicalize_all<MODE>(v, p);",
          "    canonicalize_all<MODE>(v, p);"),
         ("ntt_kernels.hip", "    if constexpr (!EARLY_HEAD) inverse_cross_head<LOGS, MODE>(cross_head, tid, cross);\n    inverse_cross_stages<LOGS, MODE, INPUT_STAGES>(v, tid, cross, mod.p, cross_head);\n    TwiddleWords head[1];",
          "    TwiddleWords head[1];")]
