DESCRIPTION = "Bfv<UInt32>: the key switch ends in key_switch_finish_kernel instead of in the 4-byte key-MAC transform's store"
EDITS = [("bfv_api.cpp",
          "    heamd::DeviceContext32 ks32{};\n    if (ks_ctx.device_context32(L + 1, ks32) != HE_OK) return hipErrorNotSupported;\n    return heamd::launch_ntt32_key_mac_inverse_finish(",
          "    heamd::DeviceContext32 ks32{};\n    if (true || ks_ctx.device_context32(L + 1, ks32) != HE_OK) return hipErrorNotSupported;\n    return heamd::launch_ntt32_key_mac_inverse_finish(")]
