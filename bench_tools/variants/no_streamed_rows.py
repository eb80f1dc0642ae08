DESCRIPTION = "round-2 N = 16384 transforms: one workgroup per row, no next-row request in flight"
EDITS = [("ntt_kernels.hip", "        if ((source == kInverseFromSlab || !inverse) && rows > compute_units())",
          "        if (false && (source == kInverseFromSlab || !inverse) && rows > compute_units())")]
