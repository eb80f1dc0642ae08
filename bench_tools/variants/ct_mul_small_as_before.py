COMPILE = ["ntt_kernels.hip", "bfv_api.cpp"]
DESCRIPTION = "ct x ct on a few ciphertexts as it was: the two operands lifted by two launches, tensor product + inverse one launch per row band (production: one launch each)"
EDITS = [("ntt_kernels.hip", "    const int count = records * record_rows <= kOneGeneration ? 0 : band_runs(ctx, record_rows, runs);", "    const int count = band_runs(ctx, record_rows, runs);"),
         ("bfv_api.cpp", """    hipError_t e = heamd::launch_lift_pair_q_to_qbsk_strided(lhs, rhs, lifted, tool.device, items, 2, 2 * L * n, 4 * ext, 0, 2 * ext,
                                                             stream, !from_source);
    if (e != hipSuccess) return e;
    if (from_source) return heamd::launch_ntt_lifted_forward(""", """    hipError_t e = heamd::launch_lift_q_to_qbsk_strided(lhs, lifted, tool.device, items, 2, 2 * L * n, 4 * ext, 0, stream, !from_source);
    if (e != hipSuccess) return e;
    e = heamd::launch_lift_q_to_qbsk_strided(rhs, lifted, tool.device, items, 2, 2 * L * n, 4 * ext, 2 * ext, stream, !from_source);
    if (e != hipSuccess) return e;
    if (from_source) return heamd::launch_ntt_lifted_forward(""")]
