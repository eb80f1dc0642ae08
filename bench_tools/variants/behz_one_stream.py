DESCRIPTION = "ct x ct: the lift and both row bands of the row-fused kernel in order on the caller's stream"
EDITS = [("bfv_api.cpp", "constexpr bool kBehzCiphertextRowsBesideLift = true;", "constexpr bool kBehzCiphertextRowsBesideLift = false;")]
