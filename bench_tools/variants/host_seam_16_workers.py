DESCRIPTION = "the pipelined host-pointer seam with 16 staging workers and 2 MiB chunks"
EDITS = [("c_api.cpp", "constexpr size_t kStageBytes = size_t(4) << 20;", "constexpr size_t kStageBytes = size_t(2) << 20;"),
         ("c_api.cpp", "constexpr size_t kStageWorkers = 8, kStageKept = 16;", "constexpr size_t kStageWorkers = 16, kStageKept = 16;")]
