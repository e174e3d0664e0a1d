// Error reporting and per-kernel hipEvent timing for libcrfp_hip.so.
#include "crfp_common.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace crfp {

static thread_local char g_err[512] = "no error";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

struct ProfEntry {
    const char* name;
    hipEvent_t start, stop;
    double bytes, flops;
};

static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfEntry> g_entries;
static std::vector<hipEvent_t> g_free_events;

bool prof_enabled() { return g_prof_on; }

static hipEvent_t get_event() {
    if (!g_free_events.empty()) {
        hipEvent_t e = g_free_events.back();
        g_free_events.pop_back();
        return e;
    }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}

ProfScope::ProfScope(const char* name, hipStream_t s, double bytes, double flops) : s_(s), slot_(-1) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfEntry e{name, get_event(), get_event(), bytes, flops};
    hipEventRecord(e.start, s);
    slot_ = (int)g_entries.size();
    g_entries.push_back(e);
}

ProfScope::~ProfScope() {
    if (slot_ < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    hipEventRecord(g_entries[slot_].stop, s_);
}

}  // namespace crfp

using namespace crfp;

extern "C" {

int crfp_version(void) { return CRFP_VERSION; }

const char* crfp_last_error_string(void) { return g_err; }

int crfp_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    return 0;
}

int crfp_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& e : g_entries) {
        hipEventSynchronize(e.stop);
        g_free_events.push_back(e.start);
        g_free_events.push_back(e.stop);
    }
    g_entries.clear();
    return 0;
}

int crfp_prof_report(crfp_prof_record* out, int cap) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    std::map<std::string, crfp_prof_record> agg;
    std::vector<std::string> order;
    for (auto& e : g_entries) {
        hipEventSynchronize(e.stop);
        float ms = 0.0f;
        hipEventElapsedTime(&ms, e.start, e.stop);
        auto it = agg.find(e.name);
        if (it == agg.end()) {
            crfp_prof_record r;
            memset(&r, 0, sizeof(r));
            strncpy(r.name, e.name, sizeof(r.name) - 1);
            it = agg.emplace(e.name, r).first;
            order.push_back(e.name);
        }
        it->second.launches += 1;
        it->second.total_ms += ms;
        it->second.bytes += e.bytes;
        it->second.flops += e.flops;
    }
    int n = 0;
    for (auto& k : order) {
        if (n < cap && out) out[n] = agg[k];
        ++n;
    }
    return n;
}

}  // extern "C"
