// Shared host-side helpers of libake_hip.so: error reporting across the C ABI and the
// hipEvent kernel timer used by bench.py's roofline leg.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/ake_hip.h"

namespace ake {

void set_error(const char* fmt, ...);

// Diagnostic switches (kernel A/B selection, phase ablation "timing only, wrong results", in-kernel cycle stamps, debug prints) are read
// from AKE_* environment variables ONLY in a diagnostic build (`AKE_DIAG=1 csrc/build.sh` -> -DAKE_DIAG).  The shipped library never looks
// at the environment: an inherited variable cannot change the results or the precision of a drop-in library;
// precision is part of the API (ake_pcnet_config::precision), the CQT engine of ake_cqt_config.  ake_build_has_diag() tells which one is loaded.
#ifdef AKE_DIAG
inline const char* diag_env(const char* name) { return std::getenv(name); }
#else
inline const char* diag_env(const char*) { return nullptr; }
#endif

#define AKE_HIP_CHECK(expr)                                                                  \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            ake::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return AKE_ERR_HIP;                                                              \
        }                                                                                    \
    } while (0)

#define AKE_REQUIRE(cond, code, ...)   \
    do {                               \
        if (!(cond)) {                 \
            ake::set_error(__VA_ARGS__); \
            return (code);             \
        }                              \
    } while (0)

// ---- kernel timer -------------------------------------------------------------------------
// When enabled, every launch site brackets its kernel with two hipEvents recorded on the
// launch stream.  Elapsed times are summed per kernel name by collect().
struct ProfScope {
    ProfScope(const char* name, hipStream_t stream);
    ~ProfScope();
    int slot;
    hipStream_t stream;
};
bool prof_active();

// One-time setup that must happen once PER DEVICE (hipFuncSetAttribute applies to the current device only; a handle may be
// recreated on another GPU of the same process).  Racing threads at worst repeat an idempotent call.
struct DeviceOnce {
    std::atomic<uint64_t> done{0};
    static int dev() { int d = 0; (void)hipGetDevice(&d); return d & 63; }
    bool need() const { return !((done.load(std::memory_order_acquire) >> dev()) & 1); }
    void mark() { done.fetch_or(1ull << dev(), std::memory_order_release); }
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Bump allocator over the caller's workspace.
struct Carver {
    char* base;
    size_t off;
    size_t cap;   // 0 = size query only
    explicit Carver(void* b, size_t c) : base(static_cast<char*>(b)), off(0), cap(c) {}
    template <typename T>
    T* take(size_t n) {
        off = align_up(off, 256);
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};

}  // namespace ake
