DESCRIPTION = ("the Galois key switch and the expand step end in key_switch_finish_kernel (rounds 2-3) instead of in the key-MAC "
               "transform's store")
EDITS = [("bfv_api.cpp", "    const bool fused_end = heamd::ntt_key_mac_finish_supported(ks, L, smallest_run * group_size);", "    const bool fused_end = false;")]
