DESCRIPTION = "ct x ct: the first of the two parts of the batch is 5/8 of it (its floor runs beside the second part's Bsk band; the second part's floor is exposed)"
EDITS = [("bfv_api.cpp", "constexpr size_t kBehzFirstPartEighths = 4;", "constexpr size_t kBehzFirstPartEighths = 5;")]
