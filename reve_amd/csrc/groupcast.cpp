#include "groupcast.h"

namespace reve {

std::string broadcast_blob(const BcastApi& api, const std::vector<int>& devs, const std::vector<void*>& ptrs, size_t bytes)
{
    const int n = (int)devs.size();
    if (n < 1 || ptrs.size() != devs.size() || bytes == 0) return "broadcast: bad arguments";
    for (void* p : ptrs)
        if (!p) return "broadcast: null device pointer";
    std::vector<void*> comms(n, nullptr);
    int rc = api.comm_init_all(comms.data(), n, devs.data());
    if (rc != 0) {
        // (a failed ncclCommInitAll hands back no communicator; anything it did leave behind is released all the same)
        for (void* c : comms)
            if (c) (void)api.comm_destroy(c);
        return std::string("ncclCommInitAll: ") + api.error_string(rc);
    }
    std::string e;
    std::vector<void*> streams(n, nullptr);
    for (int i = 0; i < n && e.empty(); ++i)
        if (api.set_device(devs[i]) != 0 || api.stream_create(&streams[i]) != 0) {
            streams[i] = nullptr;
            e = "hipStreamCreate for the broadcast failed";
        }
    bool enqueued = false;
    if (e.empty()) {
        rc = api.group_start();
        if (rc != 0) {
            e = std::string("ncclGroupStart: ") + api.error_string(rc);
        } else {
            for (int i = 0; i < n && rc == 0; ++i) {
                rc = api.broadcast(ptrs[0], ptrs[i], bytes, kNcclUint8, 0, comms[i], streams[i]);
                enqueued = true;
            }
            const int rc2 = api.group_end();          // (always: an open group would poison the thread's next collective)
            if (rc == 0) rc = rc2;
            if (rc != 0) e = std::string("ncclBroadcast: ") + api.error_string(rc);
        }
    }
    for (int i = 0; i < n; ++i)
        if (streams[i]) {
            (void)api.set_device(devs[i]);
            // (after a failure too: nothing may still be running on a stream or a communicator about to be destroyed)
            if (enqueued && api.stream_sync(streams[i]) != 0 && e.empty()) e = "broadcast stream failed";
            (void)api.stream_destroy(streams[i]);
        }
    for (void* c : comms)
        if (c) (void)api.comm_destroy(c);
    return e;
}

}  // namespace reve
