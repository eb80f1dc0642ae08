COMPILE = ["ntt_kernels.hip", "behz_kernels.hip"]
DESCRIPTION = ("per-transpose LDS padding rules in the limb-wise inverse kernels too (they keep the common rule: the rules' lane-base "
               "address words cost registers), with rules assigned to the top-partial order's transposes by analogy with the forward's")
EDITS = [
    ("ntt_rows.hpp", "    exchange<LOGN, LOGE, LO_FROM, W_FROM, LO_TO, LOGE, ROWS, !is_split(MODE) || (ROWS >= 3 && kWideGroupPerTransposeLds)>(v, tid, lds);",
     "    exchange<LOGN, LOGE, LO_FROM, W_FROM, LO_TO, LOGE, ROWS, true>(v, tid, lds);"),
    ("ntt_common.hpp", "    if constexpr (LOGN == 13 && LOGE == 3) return low == 7 ? 1 : low == 4 ? 2 : low == 1 ? 3 : 0;",
     "    constexpr int high = LO_A < LO_B ? LO_B : LO_A;\n    if constexpr (LOGN == 13 && LOGE == 3 && (low == 6 || low == 3 || (low == 0 && high == 3))) return low == 6 ? 1 : low == 3 ? 2 : 3;\n    if constexpr (LOGN == 13 && LOGE == 3) return low == 7 ? 1 : low == 4 ? 2 : low == 1 ? 3 : 0;"),
]
