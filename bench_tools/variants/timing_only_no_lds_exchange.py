COMPILE = ["ntt_kernels.hip"]
DESCRIPTION = "TIMING ONLY (wrong results): the transposition exchanges between the passes of the 8-words-per-lane kernels skipped (rows of one or two) -- what the LDS round trips and their fences cost"
EDITS = [("ntt_rows.hpp", "    constexpr int TILES = kGroupTiles<ROWS>;\n    constexpr uint32_t TILE_WORDS = lds_words(1u << LOGN);", "    constexpr int TILES = kGroupTiles<ROWS>;\n    if constexpr (ROWS < 3) { if (lds == nullptr) v[0][0] = tid; return; }\n    constexpr uint32_t TILE_WORDS = lds_words(1u << LOGN);")]
