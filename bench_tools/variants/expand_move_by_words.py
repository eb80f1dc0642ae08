COMPILE = ["galois_kernels.hip"]
DESCRIPTION = "the expansion's leaf moves and parent gathers one word per lane with the item found by division, as until round 6 (production: a workgroup inside one ciphertext, 16 bytes per lane)"
EDITS = [("galois_kernels.hip", "constexpr bool kExpandMovePairs = true;", "constexpr bool kExpandMovePairs = false;")]
