DESCRIPTION = "ct x ct: the lift reduces its Bsk rows to [0, p) (three conditional subtracts per word) instead of leaving them below 5p for the Bsk band's fold butterflies"
EDITS = [("behz_kernels.hip", "constexpr bool kBehzLazyLiftedRows = true;", "constexpr bool kBehzLazyLiftedRows = false;")]
