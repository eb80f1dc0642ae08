DESCRIPTION = "the host-pointer NTT seam as one blocking copy - kernel - copy from pageable memory (rounds 1-4)"
EDITS = [("c_api.cpp", "    if (chunks < 2) {  // a polynomial or two", "    if (true) {  // a polynomial or two")]
