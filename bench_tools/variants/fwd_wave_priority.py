DESCRIPTION = ("the forward transform raises the issue priority of the workgroup that sits in the upper wave slots of its SIMDs "
               "(s_setprio by HW_ID.wave_id), as inv_wave_priority does for the inverse")
EDITS = [("ntt_kernels.hip", """    const uint32_t tid = threadIdx.x;
    uint32_t record, within;
    size_t rows[ROWS];
    if constexpr (SPREAD != kSourceSlab && SPREAD != kSourceRows) {""",
          """    const uint32_t tid = threadIdx.x;
    {
        uint32_t hw_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        if ((hw_id & 4u) != 0) __builtin_amdgcn_s_setprio(3);
    }
    uint32_t record, within;
    size_t rows[ROWS];
    if constexpr (SPREAD != kSourceSlab && SPREAD != kSourceRows) {""")]
