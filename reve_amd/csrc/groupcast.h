// The one collective of the path: the packed weights, uploaded to devices[0], reach the other GPUs of a group
// (reve_create_group: the binary's `-g 0,1,...`) by ONE ncclBroadcast over xGMI.  broadcast_blob() is the whole exchange as a pure
// function over device ordinals, device pointers, a byte count and a table of the calls it makes — RCCL's and the four stream
// calls around them — so that its sequence and every unwinding path can be driven on a CPU by a recording table
// (san/fake_rccl.cpp, tests/test_sanitizers.py) before it ever meets n > 1 GPUs.  The product table is rccl_api.cpp.
#pragma once
#include <cstddef>
#include <functional>
#include <string>
#include <vector>

namespace reve {

struct BcastApi {
    // RCCL (NCCL's names and return convention: 0 = ncclSuccess)
    std::function<int(void** comms, int n, const int* devs)> comm_init_all;
    std::function<int(void* comm)> comm_destroy;
    std::function<int()> group_start, group_end;
    std::function<int(const void* send, void* recv, size_t count, int dtype, int root, void* comm, void* stream)> broadcast;
    std::function<const char*(int)> error_string;
    // HIP (0 = hipSuccess): the stream each rank's broadcast is enqueued on, on that rank's device
    std::function<int(int device)> set_device;
    std::function<int(void** stream)> stream_create;
    std::function<int(void* stream)> stream_sync, stream_destroy;
};

constexpr int kNcclUint8 = 1;       // ncclUint8 (nccl.h: ncclInt8 = 0, ncclUint8 = 1)

// ptrs[i] on device devs[i], `bytes` each; ptrs[0] holds the blob, afterwards all do (in place on the root).  One communicator
// per device from one comm_init_all, one group_start / group_end around n broadcasts (count = bytes, ncclUint8, root 0, comm i on
// stream i), then every stream is synchronised.  Whatever fails, every stream created is destroyed and every communicator
// created is destroyed, once.  Returns "" or the first failure's text.
std::string broadcast_blob(const BcastApi& api, const std::vector<int>& devs, const std::vector<void*>& ptrs, size_t bytes);

// The table of the running system: librccl (dlopen'ed on first use, so a single-GPU caller never needs it) + the HIP runtime.
// nullptr with `err` set when librccl cannot be loaded or lacks an entry point (sticky).  rccl_api.cpp; san/fake_rccl.cpp in the
// CPU sanitizer builds.
const BcastApi* system_bcast_api(std::string& err);

}  // namespace reve
