DESCRIPTION = "ct x ct: the Bsk band and the floor in four parts of the batch"
EDITS = [("bfv_api.cpp", "constexpr size_t kBehzFloorParts = 2;", "constexpr size_t kBehzFloorParts = 4;")]
