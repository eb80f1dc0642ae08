DESCRIPTION = "ct x ct: the Bsk band and the floor over the whole batch in order (no floor of one part beside the next part's Bsk band)"
EDITS = [("bfv_api.cpp", "constexpr size_t kBehzFloorParts = 2;", "constexpr size_t kBehzFloorParts = 1;")]
