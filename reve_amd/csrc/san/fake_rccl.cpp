// CPU sanitizer builds only: the BcastApi (groupcast.h) of a machine with no GPU — a RECORDING table.  Every call is appended to a
// log the harness reads; "communicators" and "streams" are heap cookies, so AddressSanitizer sees a double destroy or a leak; the
// "broadcast" is performed at stream_sync (memcpy root -> rank: the fake device blobs are host memory), so that a sequence which
// forgets to synchronise leaves the other ranks' blobs empty.  Failures are injected by call name and ordinal:
//   REVE_FAKE_RCCL_FAIL="<call>[:<k>]"   e.g. comm_init_all, broadcast:2 (the third), group_end, stream_sync:1, stream_create:3, group_start
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../groupcast.h"

namespace reve {

namespace {
struct Cookie { char kind; int index; bool live; };
struct Pending { const void* send; void* recv; size_t n; void* stream; };
struct Fake {
    std::mutex mu;
    std::vector<std::string> log;
    std::map<std::string, int> calls;
    std::vector<Cookie*> cookies;
    std::vector<Pending> pending;
    int current_device = -1;
    BcastApi api;
    bool fails(const std::string& call)
    {
        const int k = calls[call]++;
        const char* e = std::getenv("REVE_FAKE_RCCL_FAIL");
        if (!e) return false;
        const std::string spec = e;
        const size_t colon = spec.find(':');
        return spec.substr(0, colon) == call && (colon == std::string::npos ? 0 : std::atoi(spec.c_str() + colon + 1)) == k;
    }
    Cookie* make(char kind, int index) { Cookie* c = new Cookie{kind, index, true}; cookies.push_back(c); return c; }
    Fake()
    {
        api.comm_init_all = [this](void** comms, int n, const int* devs) {
            std::lock_guard<std::mutex> lk(mu);
            std::string l = "comm_init_all " + std::to_string(n) + " devs";
            for (int i = 0; i < n; ++i) l += " " + std::to_string(devs[i]);
            log.push_back(l);
            if (fails("comm_init_all")) return 3;          // ncclInternalError
            for (int i = 0; i < n; ++i) comms[i] = make('c', i);
            return 0;
        };
        api.comm_destroy = [this](void* c) {
            std::lock_guard<std::mutex> lk(mu);
            Cookie* k = (Cookie*)c;
            log.push_back(std::string("comm_destroy ") + (k->kind == 'c' && k->live ? "comm" : "BAD") + std::to_string(k->index));
            k->live = false;
            return 0;
        };
        api.group_start = [this] { std::lock_guard<std::mutex> lk(mu); log.push_back("group_start"); return fails("group_start") ? 3 : 0; };
        api.group_end = [this] { std::lock_guard<std::mutex> lk(mu); log.push_back("group_end"); return fails("group_end") ? 1 : 0; };      // ncclUnhandledCudaError
        api.broadcast = [this](const void* send, void* recv, size_t n, int dtype, int root, void* comm, void* stream) {
            std::lock_guard<std::mutex> lk(mu);
            Cookie* c = (Cookie*)comm;
            Cookie* s = (Cookie*)stream;
            log.push_back("broadcast count " + std::to_string(n) + " dtype " + std::to_string(dtype) + " root " + std::to_string(root) + " comm" +
                          std::to_string(c->index) + " stream" + std::to_string(s->index) + (c->live && s->live && c->kind == 'c' && s->kind == 's' ? "" : " BAD") +
                          (send == recv ? " in-place" : ""));
            if (fails("broadcast")) return 5;              // ncclInvalidUsage
            pending.push_back({send, recv, n, stream});
            return 0;
        };
        api.error_string = [](int rc) { return rc == 1 ? "unhandled cuda error" : rc == 3 ? "internal error" : rc == 5 ? "invalid usage" : "fake error"; };
        api.set_device = [this](int d) { std::lock_guard<std::mutex> lk(mu); current_device = d; return 0; };
        api.stream_create = [this](void** s) {
            std::lock_guard<std::mutex> lk(mu);
            log.push_back("stream_create on device " + std::to_string(current_device));
            if (fails("stream_create")) return 2;
            int idx = 0;
            for (Cookie* c : cookies) idx += c->kind == 's' && c->live;
            *s = make('s', idx);
            return 0;
        };
        api.stream_sync = [this](void* s) {
            std::lock_guard<std::mutex> lk(mu);
            Cookie* k = (Cookie*)s;
            log.push_back(std::string("stream_sync ") + (k->kind == 's' && k->live ? "stream" : "BAD") + std::to_string(k->index) + " on device " + std::to_string(current_device));
            if (fails("stream_sync")) return 719;          // hipErrorLaunchFailure
            for (Pending& p : pending)
                if (p.stream == s && p.recv != p.send && p.n) std::memcpy(p.recv, p.send, p.n);
            return 0;
        };
        api.stream_destroy = [this](void* s) {
            std::lock_guard<std::mutex> lk(mu);
            Cookie* k = (Cookie*)s;
            log.push_back(std::string("stream_destroy ") + (k->kind == 's' && k->live ? "stream" : "BAD") + std::to_string(k->index));
            k->live = false;
            return 0;
        };
    }
};
Fake& fake() { static Fake* f = new Fake; return *f; }
}  // namespace

const BcastApi* system_bcast_api(std::string& err)
{
    if (const char* e = std::getenv("REVE_FAKE_RCCL_MISSING"); e && e[0] == '1') { err = "cannot load librccl: injected"; return nullptr; }
    err.clear();
    return &fake().api;
}

// for the harness: the log so far (and a reset between scenarios); cookies still live = leaked comms / streams
std::vector<std::string> fake_rccl_take_log(int* live_comms, int* live_streams)
{
    Fake& f = fake();
    std::lock_guard<std::mutex> lk(f.mu);
    int c = 0, s = 0;
    for (Cookie* k : f.cookies) { c += k->kind == 'c' && k->live; s += k->kind == 's' && k->live; delete k; }
    f.cookies.clear();
    f.pending.clear();
    f.calls.clear();
    if (live_comms) *live_comms = c;
    if (live_streams) *live_streams = s;
    std::vector<std::string> out;
    out.swap(f.log);
    return out;
}

}  // namespace reve
