DESCRIPTION = ("ct x ct without the row-fused BEHZ kernel (rounds 1-4): forward transforms of the lifted [Q, Bsk] records written "
               "to HBM (two launches), tensor product fused into the inverse transform's load (two launches)")
EDITS = [("bfv_api.cpp", "constexpr bool kBehzRowsFused = true;", "constexpr bool kBehzRowsFused = false;")]
