// Error reporting and device probing for libsequoia_hip.
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>

#include "../../include/sequoia_hip.h"
#include "sq_common.h"

static thread_local char g_err[512] = "";

void sq_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* sq_last_error(void) { return g_err; }

extern "C" int sq_version(void) { return 100; }

extern "C" int sq_device_ok(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return 0; }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

// ------------------------------------------------------------------------------------
// HIP-event kernel timing, used by bench.py's roofline leg: every instrumented launch is
// bracketed by two events on the launch stream; sq_prof_report aggregates by kernel name.
// Off by default (no events are recorded in the timed region of the headline number).
// ------------------------------------------------------------------------------------
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {
struct ProfRec { hipEvent_t a, b; std::string name; double flops, bytes; };
std::mutex g_prof_mu;
bool g_prof_on = false;
bool g_prof_markers = false;               // sq_prof_enable(2): a marker launch in front of and behind every instrumented launch
std::vector<std::string> g_marker_names;   // class id (first appearance) -> name
std::vector<ProfRec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
}  // namespace

bool sq_prof_on() { return g_prof_on; }

// A kernel that does nothing, whose GRID SIZE carries a number: a profiler that lists dispatches in order (rocprofv3 --kernel-trace
// --pmc) then shows which dispatches belong to which instrumented launch without knowing any kernel symbol or launch geometry:
// (id + 2) blocks of 64 threads open class `id`, one block closes it (bench.py measure_traffic_marked).
__global__ void sq_prof_marker_kernel(int* sink) { if (sink) sink[0] = (int)gridDim.x; }

int sq_prof_begin(const char* name, double flops, double bytes, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_on) return -1;
    ProfRec r{get_event(), get_event(), name, flops, bytes};
    if (!r.a || !r.b) return -1;
    if (g_prof_markers) {
        size_t id = 0;
        while (id < g_marker_names.size() && g_marker_names[id] != name) ++id;
        if (id == g_marker_names.size()) g_marker_names.push_back(name);
        hipLaunchKernelGGL(sq_prof_marker_kernel, dim3((unsigned)id + 2), dim3(64), 0, st, (int*)nullptr);
    }
    (void)hipEventRecord(r.a, st);
    g_recs.push_back(r);
    return (int)g_recs.size() - 1;
}

void sq_prof_end(int idx, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (idx < 0 || idx >= (int)g_recs.size()) return;
    (void)hipEventRecord(g_recs[idx].b, st);
    if (g_prof_markers) hipLaunchKernelGGL(sq_prof_marker_kernel, dim3(1), dim3(64), 0, st, (int*)nullptr);
}

extern "C" int sq_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    g_prof_markers = on == 2;
    return SQ_OK;
}

// JSON array of the class names the marker launches of sq_prof_enable(2) have numbered so far (index = class id).
extern "C" int sq_prof_marker_names(char* buf, size_t cap) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    std::string s = "[";
    for (size_t i = 0; i < g_marker_names.size(); ++i) s += (i ? ",\"" : "\"") + g_marker_names[i] + "\"";
    s += "]";
    SQ_REQUIRE(buf && s.size() + 1 <= cap, "prof_marker_names: buffer of %zu bytes too small (%zu needed)", cap, s.size() + 1);
    memcpy(buf, s.c_str(), s.size() + 1);
    return SQ_OK;
}

// Writes a JSON array [{"name":..,"count":..,"total_ms":..,"flops":..,"bytes":..}, ...] (flops/bytes per launch)
// into buf, clears the records.  The caller synchronises the stream(s) first.
extern "C" int sq_prof_report(char* buf, size_t cap) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    struct Agg { long count = 0; double ms = 0, flops = 0, bytes = 0; };
    std::map<std::string, Agg> agg;
    for (auto& r : g_recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            Agg& a = agg[r.name];
            a.count++; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes;      // summed; reported as the per-launch mean
        }
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_recs.clear();
    std::string s = "[";
    bool first = true;
    for (auto& kv : agg) {
        char line[384];
        snprintf(line, sizeof(line), "%s{\"name\":\"%s\",\"count\":%ld,\"total_ms\":%.6f,\"flops\":%.1f,\"bytes\":%.1f}",
                 first ? "" : ",", kv.first.c_str(), kv.second.count, kv.second.ms, kv.second.flops / kv.second.count,
                 kv.second.bytes / kv.second.count);
        s += line;
        first = false;
    }
    s += "]";
    SQ_REQUIRE(buf && s.size() + 1 <= cap, "prof_report: buffer of %zu bytes too small (%zu needed)", cap, s.size() + 1);
    memcpy(buf, s.c_str(), s.size() + 1);
    return SQ_OK;
}

namespace {
struct SideSlot { hipStream_t stream = nullptr; std::vector<hipEvent_t> events; SqSideStream view{}; };
std::mutex g_side_mu;
std::map<std::pair<int, int>, SideSlot> g_side;
}  // namespace

SqSideStream* sq_side_stream(int which, int n_events) {
    std::lock_guard<std::mutex> lk(g_side_mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    SideSlot& s = g_side[{dev, which}];
    if (!s.stream && hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
    while ((int)s.events.size() < n_events) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        s.events.push_back(e);
    }
    s.view.stream = s.stream; s.view.events = s.events.data(); s.view.n_events = (int)s.events.size();
    return &s.view;
}

bool sq_env_flag(const char* name) {
    const char* v = getenv(name);
    return v && v[0] && v[0] != '0';
}
