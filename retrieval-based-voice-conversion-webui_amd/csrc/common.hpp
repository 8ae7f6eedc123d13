// Shared host-side plumbing for librvcmi.so: error reporting across the C ABI, HIP call checking,
// device buffers and the HIP-event kernel profiler used by bench.py's roofline leg.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../include/rvcmi.h"

namespace rvcmi {

void set_error(const char* fmt, ...);

struct Error {
    int code;
};

#define RVCMI_FAIL(code_, ...)            \
    do {                                  \
        ::rvcmi::set_error(__VA_ARGS__);  \
        throw ::rvcmi::Error{(code_)};    \
    } while (0)

#define HIP_CHECK(expr)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            RVCMI_FAIL(RVCMI_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),     \
                       __FILE__, __LINE__);                                                      \
    } while (0)

// Every extern "C" body runs inside this so that nothing throws across the ABI.
template <typename F>
int guarded(F&& f) {
    try {
        f();
        return RVCMI_OK;
    } catch (const Error& e) {
        return e.code;
    } catch (const std::exception& e) {
        set_error("exception: %s", e.what());
        return RVCMI_ERR_INVALID;
    } catch (...) {
        set_error("unknown exception");
        return RVCMI_ERR_INVALID;
    }
}

// Makes `device` current for the scope and restores the caller's device afterwards (the thread's current device is torch's too).
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int device) {
        HIP_CHECK(hipGetDevice(&prev));
        if (prev != device) HIP_CHECK(hipSetDevice(device));
        else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    void alloc(size_t n) {
        free();
        if (n == 0) return;
        HIP_CHECK(hipMalloc(&p, n));
        bytes = n;
    }
    void free() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <typename T>
    T* as() const {
        return reinterpret_cast<T*>(p);
    }
    ~DevBuf() { free(); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) {
        o.p = nullptr;
        o.bytes = 0;
    }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) {
            free();
            p = o.p;
            bytes = o.bytes;
            o.p = nullptr;
            o.bytes = 0;
        }
        return *this;
    }
};

// Dev / test options of a handle: a small key -> number table, filled ONCE at handle creation from the environment
// (RVCMI_<KEY>) and changed afterwards only through rvcmi_{nsf,front,ivf}_set_option.  The forward / search paths read the
// table; nothing on them calls getenv.  Keys are listed next to the handles that honour them.
struct Options {
    std::map<std::string, double> v;
    std::set<std::string> known;  // the keys this handle honours (everything ever passed to load_env)
    void load_env(std::initializer_list<const char*> keys) {
        for (const char* k : keys) {
            known.insert(k);
            const std::string e = std::string("RVCMI_") + k;
            if (const char* sv = getenv(e.c_str())) v[k] = atof(sv);
        }
    }
    bool has(const char* k) const { return v.find(k) != v.end(); }
    double get(const char* k, double dflt) const {
        auto it = v.find(k);
        return it == v.end() ? dflt : it->second;
    }
    int geti(const char* k, int dflt) const { return (int)get(k, (double)dflt); }
    bool on(const char* k) const { return get(k, 0.0) != 0.0; }
    // false: not a key of this handle (a typo would otherwise pin nothing and a test would silently run the default path)
    bool set(const char* k, double val) {
        if (!known.count(k)) return false;
        if (val != val) v.erase(k);  // NaN: back to the default
        else v[k] = val;
        return true;
    }
};

// HIP-event bracketed launches.  Disabled: zero overhead.  Enabled: one event pair per launch on
// the launch stream; read() resolves elapsed times and folds them into per-name statistics.
class Profiler {
   public:
    bool enabled = false;
    struct Pending {
        int stat;
        hipEvent_t a, b;
    };
    std::vector<rvcmi_kernel_stat> stats;
    std::map<std::string, int> index;
    std::vector<Pending> pending;

    int stat_id(const char* name) {
        auto it = index.find(name);
        if (it != index.end()) return it->second;
        rvcmi_kernel_stat s;
        memset(&s, 0, sizeof(s));
        strncpy(s.name, name, sizeof(s.name) - 1);
        stats.push_back(s);
        index[name] = (int)stats.size() - 1;
        return (int)stats.size() - 1;
    }
    template <typename F>
    void launch(const char* name, double flops, double bytes, hipStream_t st, F&& f) {
        if (!enabled) {
            f();
            return;
        }
        int id = stat_id(name);
        Pending p;
        p.stat = id;
        HIP_CHECK(hipEventCreate(&p.a));
        HIP_CHECK(hipEventCreate(&p.b));
        HIP_CHECK(hipEventRecord(p.a, st));
        f();
        HIP_CHECK(hipEventRecord(p.b, st));
        stats[id].launches += 1;
        stats[id].flops += flops;
        stats[id].bytes += bytes;
        pending.push_back(p);
    }
    void resolve() {
        for (auto& p : pending) {
            HIP_CHECK(hipEventSynchronize(p.b));
            float ms = 0.f;
            HIP_CHECK(hipEventElapsedTime(&ms, p.a, p.b));
            stats[p.stat].ms += ms;
            (void)hipEventDestroy(p.a);
            (void)hipEventDestroy(p.b);
        }
        pending.clear();
    }
    int read(rvcmi_kernel_stat* out, int capacity, int* n, int reset) {
        resolve();
        int m = (int)stats.size();
        if (n) *n = m;
        for (int i = 0; i < m && i < capacity; ++i) out[i] = stats[i];
        if (reset) {
            stats.clear();
            index.clear();
        }
        return 0;
    }
};

}  // namespace rvcmi
