DESCRIPTION = ("the inverse transform raises the issue priority of the workgroup that sits in the upper wave slots of its SIMDs "
               "(s_setprio by HW_ID.wave_id): the two workgroups of a CU drift out of phase, one computes while the other waits")
EDITS = [("ntt_kernels.hip", """    const uint32_t tid = threadIdx.x;
    uint32_t record, within;
    size_t rows[ROWS];
    if constexpr (!FROM_SLAB) {
        // records (item, c) of one item read the same source rows""",
          """    const uint32_t tid = threadIdx.x;
    {
        uint32_t hw_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        if ((hw_id & 4u) != 0) __builtin_amdgcn_s_setprio(3);
    }
    uint32_t record, within;
    size_t rows[ROWS];
    if constexpr (!FROM_SLAB) {
        // records (item, c) of one item read the same source rows""")]
