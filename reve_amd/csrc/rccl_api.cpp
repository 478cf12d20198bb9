// The product's BcastApi (groupcast.h): librccl through dlopen + the HIP runtime's stream calls.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <mutex>

#include "groupcast.h"

namespace reve {

namespace {
struct Rccl {
    void* lib = nullptr;
    int (*CommInitAll)(void**, int, const int*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;         // why the library is unusable ("" = loaded and complete); set once
    std::once_flag once;
    BcastApi api;
    // thread-safe and sticky: two threads creating groups at once load it once, and a librccl that lacks a symbol stays an
    // error on every later call
    const std::string& load()
    {
        std::call_once(once, [this] {
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
                if ((lib = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
            if (!lib) { const char* why = dlerror(); error = std::string("cannot load librccl: ") + (why ? why : "unknown error"); return; }
            auto sym = [&](const char* n) { return dlsym(lib, n); };
            CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
            CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
            GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
            GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
            Broadcast = (decltype(Broadcast))sym("ncclBroadcast");
            GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
            if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Broadcast || !GetErrorString) {
                error = "librccl lacks the NCCL entry points";
                dlclose(lib);
                lib = nullptr;
                return;
            }
            api.comm_init_all = CommInitAll;
            api.comm_destroy = CommDestroy;
            api.group_start = GroupStart;
            api.group_end = GroupEnd;
            api.broadcast = [this](const void* s, void* r, size_t n, int t, int root, void* c, void* st) { return Broadcast(s, r, n, t, root, c, (hipStream_t)st); };
            api.error_string = GetErrorString;
            api.set_device = [](int d) { return (int)hipSetDevice(d); };
            api.stream_create = [](void** s) { hipStream_t st = nullptr; const int rc = (int)hipStreamCreateWithFlags(&st, hipStreamNonBlocking); *s = st; return rc; };
            api.stream_sync = [](void* s) { return (int)hipStreamSynchronize((hipStream_t)s); };
            api.stream_destroy = [](void* s) { return (int)hipStreamDestroy((hipStream_t)s); };
        });
        return error;
    }
};
}  // namespace

const BcastApi* system_bcast_api(std::string& err)
{
    static Rccl* r = new Rccl;          // (never destroyed: contexts may be created during process exit handlers' lifetime)
    err = r->load();
    return err.empty() ? &r->api : nullptr;
}

}  // namespace reve
