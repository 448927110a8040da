// Error reporting + hipEvent kernel timer (see common.h).
#include "common.h"

#include <cstring>
#include <map>
#include <mutex>

namespace ake {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

namespace {
struct Pending {
    int name_id;
    hipEvent_t start, stop;
};
struct Entry {
    std::string name;
    double total_ms = 0;
    int64_t launches = 0;
};
std::mutex g_mu;
bool g_on = false;
std::string g_filter;
std::vector<Pending> g_pending;
std::vector<hipEvent_t> g_free_events;
std::vector<Entry> g_entries;
std::map<std::string, int> g_index;

hipEvent_t get_event() {
    if (!g_free_events.empty()) {
        hipEvent_t e = g_free_events.back();
        g_free_events.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

bool prof_active() { return g_on; }

ProfScope::ProfScope(const char* name, hipStream_t s) : slot(-1), stream(s) {
    if (!g_on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_filter.empty() && std::strstr(name, g_filter.c_str()) == nullptr) return;
    auto it = g_index.find(name);
    int id;
    if (it == g_index.end()) {
        id = static_cast<int>(g_entries.size());
        g_entries.push_back(Entry{name});
        g_index[name] = id;
    } else {
        id = it->second;
    }
    Pending p{id, get_event(), get_event()};
    (void)hipEventRecord(p.start, stream);
    slot = static_cast<int>(g_pending.size());
    g_pending.push_back(p);
}

ProfScope::~ProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    (void)hipEventRecord(g_pending[slot].stop, stream);
}

}  // namespace ake

using namespace ake;

extern "C" {

int ake_version(void) { return 101; }
int ake_build_has_diag(void) {
#ifdef AKE_DIAG
    return 1;
#else
    return 0;
#endif
}

const char* ake_last_error(void) { return g_err; }

int ake_prof_enable(const char* name_filter, int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_filter = name_filter ? name_filter : "";
    g_on = on != 0;
    return AKE_OK;
}

int ake_prof_collect(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& p : g_pending) {
        AKE_HIP_CHECK(hipEventSynchronize(p.stop));
        float ms = 0.f;
        AKE_HIP_CHECK(hipEventElapsedTime(&ms, p.start, p.stop));
        g_entries[p.name_id].total_ms += ms;
        g_entries[p.name_id].launches += 1;
        g_free_events.push_back(p.start);
        g_free_events.push_back(p.stop);
    }
    g_pending.clear();
    return AKE_OK;
}

int ake_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& p : g_pending) {
        g_free_events.push_back(p.start);
        g_free_events.push_back(p.stop);
    }
    g_pending.clear();
    g_entries.clear();
    g_index.clear();
    return AKE_OK;
}

int ake_prof_num_entries(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return static_cast<int>(g_entries.size());
}

int ake_prof_entry(int index, const char** kernel_name, double* total_ms, int64_t* launches) {
    std::lock_guard<std::mutex> lk(g_mu);
    AKE_REQUIRE(index >= 0 && index < static_cast<int>(g_entries.size()), AKE_ERR_INVALID, "prof entry %d out of range", index);
    if (kernel_name) *kernel_name = g_entries[index].name.c_str();
    if (total_ms) *total_ms = g_entries[index].total_ms;
    if (launches) *launches = g_entries[index].launches;
    return AKE_OK;
}

}  // extern "C"
