DESCRIPTION = ("Bfv<UInt32> lift / floor with one 4-byte word per lane (rounds 2-3) instead of four (one 16-byte access per lane)")
EDITS = [
    ("rns_kernels.hip", "            if (quad_aligned(in) && quad_aligned(out) && layout.in_item_stride % 4 == 0",
     "            if (false && quad_aligned(in) && quad_aligned(out) && layout.in_item_stride % 4 == 0"),
    ("rns_kernels.hip", "            if (quad_aligned(in) && quad_aligned(out) && tool.log_degree >= 2) {",
     "            if (false && quad_aligned(in) && quad_aligned(out) && tool.log_degree >= 2) {"),
]
