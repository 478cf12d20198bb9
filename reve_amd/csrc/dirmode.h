// Directory mode: the file contract Video::upscale_segment relies on
// (reve-shared/src/lib.rs:130-147): every frame image in `in_dir` is upscaled to
// `out_dir/<same stem>.png`; one progress callback per finished frame, in name order.
#pragma once
#include <string>
#include <vector>

#include "../../include/reve_hip.h"
#include "engine.h"

namespace reve {
// engs: one engine per GPU (same scale); frame f of the sorted directory goes to engs[f mod G]
int upscale_dir(const std::vector<Engine*>& engs, const std::string& in_dir, const std::string& out_dir, reve_progress_cb cb,
                void* user, std::string& err);
// CPUs the process may use: affinity mask and control-group CPU quota (dirmode.cpp).  `root` is prefixed to /proc/self/cgroup and
// /sys/fs/cgroup (tests point it at a directory of their own).
int effective_cpus(const std::string& root = "");

int upscale_file(Engine& eng, const std::string& in_path, const std::string& out_path, std::string& err);
}  // namespace reve
