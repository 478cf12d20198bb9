// roctx ranges around the stages of the path (SURVEY.md §5 tracing row): upload / chain / download enqueue in the ring,
// decode / encode / GPU wait in the host pipeline.  With `rocprofv3 --marker-trace --kernel-trace --memory-copy-trace` one
// trace then shows the host stages next to the kernels and copies they overlap.  The roctx library is looked up at first use
// — only if it is already in the process (the profiler loads it) or REVE_ROCTX=1 asks for it — so a normal run pays one
// branch per range and has no dependency on the profiler's libraries.
#pragma once
#include <dlfcn.h>

#include <cstdlib>

namespace reve {
struct RoctxApi {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    RoctxApi()
    {
        const bool want = std::getenv("REVE_ROCTX") && std::getenv("REVE_ROCTX")[0] == '1';
        void* lib = nullptr;
        for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (!lib && want) lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return;
        push = (int (*)(const char*))dlsym(lib, "roctxRangePushA");
        pop = (int (*)())dlsym(lib, "roctxRangePop");
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
};
inline RoctxApi& roctx_api() { static RoctxApi a; return a; }
struct TraceRange {
    bool on;
    explicit TraceRange(const char* name) : on(roctx_api().push != nullptr) { if (on) roctx_api().push(name); }
    ~TraceRange() { end(); }
    void end() { if (on) { roctx_api().pop(); on = false; } }
    TraceRange(const TraceRange&) = delete;
    TraceRange& operator=(const TraceRange&) = delete;
};
}  // namespace reve
