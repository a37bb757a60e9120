// state.hip — the process-wide state libicpmi.so owns, all of it here and all of it declared in include/icpmi.h:
//
//  * options: the ICPMI_<NAME> environment switches (experiments and tests; none is needed in production) are read
//    ONCE, at the first call that asks for one, and kept; icpmi_set_option changes one afterwards.  No launch path
//    calls getenv.
//  * side streams: up to ICPMI_SIDE_STREAMS streams per device with a fork event and one join event each, made on
//    first use (option ICP2_SIDE = 0: never), used by launchers that overlap independent launches of one call (the
//    wide-cloud launch of the fused ICP beside its second stage).  A launcher
//    holds the lock for its whole fork / launch / join sequence, so two host threads never interleave their event
//    records; every caller stream of a device shares these streams (calls from several streams overlap their side
//    work only as far as the side streams differ).  icpmi_shutdown destroys them.
//
//  * kernel attributes: the largest dynamic-LDS size asked of a kernel on a device is remembered (dyn_lds), so that
//    hipFuncSetAttribute is called when a launch needs more than any before it, not on every launch (it cost the
//    0.2 ms single-pair path several microseconds of host time per launch).  The attribute lives in the HIP runtime;
//    this is a mirror of what was set, nothing to destroy.
//
// Nothing else in the library outlives a call: no allocations, no handles.
#include <link.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <mutex>
#include <string>

#include "common.hpp"

namespace icpmi {

static const char* const kOptionNames[] = {"ICP2_SIDE", "ICP2_FAR", "ICP2_SHAPE", "ICP2_FILTER", "ICP2_STAGES", "POLAR", "PREP_KNN", "RAYCAST",
                                           "RT_WGS", "RS_BATCH"};

struct Options {
    std::mutex mu;
    bool loaded = false;
    std::map<std::string, std::string> values;
    void load() {                                                 // under mu
        if (loaded) return;
        loaded = true;
        for (const char* n : kOptionNames) {
            const std::string env = std::string("ICPMI_") + n;
            if (const char* v = getenv(env.c_str())) values[n] = v;
        }
    }
};
static Options& options() { static Options o; return o; }

static const char* strip_prefix(const char* name) { return strncmp(name, "ICPMI_", 6) == 0 ? name + 6 : name; }

// value of option `name` (with or without the ICPMI_ prefix), or nullptr when unset.  The pointer stays valid until
// that option is set again (icpmi_set_option must not race with calls that read it).
const char* option(const char* name) {
    Options& o = options();
    std::lock_guard<std::mutex> g(o.mu);
    o.load();
    const auto it = o.values.find(strip_prefix(name));
    return it == o.values.end() ? nullptr : it->second.c_str();
}

struct LdsAttr {
    std::mutex mu;
    std::map<std::pair<const void*, int>, size_t> granted;
};
static LdsAttr& lds_attr() { static LdsAttr a; return a; }

// hipFuncAttributeMaxDynamicSharedMemorySize of kernel `fn` on the current device raised to at least `bytes`
hipError_t dyn_lds(const void* fn, size_t bytes) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    LdsAttr& a = lds_attr();
    std::lock_guard<std::mutex> g(a.mu);
    size_t& have = a.granted[std::make_pair(fn, dev)];
    if (bytes <= have) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) have = bytes;
    return e;
}

struct SideState {
    std::mutex mu;
    std::map<int, Side> per_device;
};
static SideState& side_state() { static SideState s; return s; }

SideLock::SideLock() : side(nullptr) {
    SideState& s = side_state();
    s.mu.lock();
    const char* e = option("ICP2_SIDE");
    if (e && e[0] == '0') return;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return;
    Side& sd = s.per_device[dev];
    if (sd.failed) return;
    if (!sd.fork) {
        bool ok = hipEventCreateWithFlags(&sd.fork, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; ok && i < ICPMI_SIDE_STREAMS; ++i)
            ok = hipStreamCreateWithFlags(&sd.stream[i], hipStreamNonBlocking) == hipSuccess &&
                 hipEventCreateWithFlags(&sd.join[i], hipEventDisableTiming) == hipSuccess;
        if (!ok) { sd.failed = true; return; }
    }
    side = &sd;
}
SideLock::~SideLock() { side_state().mu.unlock(); }

}  // namespace icpmi

extern "C" int icpmi_set_option(const char* name, const char* value) {
    using namespace icpmi;
    if (!name) return ICPMI_ERR_ARG;
    const char* n = strip_prefix(name);
    bool known = false;
    for (const char* k : kOptionNames) known = known || strcmp(k, n) == 0;
    if (!known) return ICPMI_ERR_ARG;
    Options& o = options();
    std::lock_guard<std::mutex> g(o.mu);
    o.load();
    if (value) o.values[n] = value; else o.values.erase(n);
    return ICPMI_OK;
}

static int count_hip_runtimes(struct dl_phdr_info* info, size_t, void* data) {
    if (info->dlpi_name && strstr(info->dlpi_name, "libamdhip64")) static_cast<std::set<std::string>*>(data)->insert(info->dlpi_name);
    return 0;
}

extern "C" int icpmi_runtime_check(char* msg, size_t msg_bytes) {
    std::set<std::string> seen;
    dl_iterate_phdr(count_hip_runtimes, &seen);
    if (msg && msg_bytes) msg[0] = 0;
    if (seen.size() <= 1) return ICPMI_OK;
    if (msg && msg_bytes) {
        std::string m = "two HIP runtimes in one process:";
        for (const std::string& p : seen) m += " " + p;
        m += " - load the framework that brings its own (import torch) BEFORE libicpmi.so";
        snprintf(msg, msg_bytes, "%s", m.c_str());
    }
    return ICPMI_ERR_HIP;
}

extern "C" int icpmi_shutdown(void) {
    using namespace icpmi;
    SideState& s = side_state();
    std::lock_guard<std::mutex> g(s.mu);
    int rc = ICPMI_OK;
    int cur = -1;
    (void)hipGetDevice(&cur);
    for (auto& kv : s.per_device) {
        Side& sd = kv.second;
        if (hipSetDevice(kv.first) != hipSuccess) { rc = ICPMI_ERR_HIP; continue; }
        for (int i = 0; i < ICPMI_SIDE_STREAMS; ++i) {
            if (sd.stream[i]) { (void)hipStreamSynchronize(sd.stream[i]); if (hipStreamDestroy(sd.stream[i]) != hipSuccess) rc = ICPMI_ERR_HIP; }
            if (sd.join[i] && hipEventDestroy(sd.join[i]) != hipSuccess) rc = ICPMI_ERR_HIP;
        }
        if (sd.fork && hipEventDestroy(sd.fork) != hipSuccess) rc = ICPMI_ERR_HIP;
    }
    s.per_device.clear();
    if (cur >= 0) (void)hipSetDevice(cur);
    return rc;
}
