DESCRIPTION = "the pipelined host-pointer seam with 4 staging workers and 8 MiB chunks"
EDITS = [("c_api.cpp", "constexpr size_t kStageBytes = size_t(4) << 20;", "constexpr size_t kStageBytes = size_t(8) << 20;"),
         ("c_api.cpp", "constexpr size_t kStageWorkers = 8, kStageKept = 16;", "constexpr size_t kStageWorkers = 4, kStageKept = 16;")]
