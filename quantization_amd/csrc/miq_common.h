// Shared host-side helpers for the miq C ABI (error text, argument checks, launch check).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <initializer_list>

#include "miq.h"

namespace miq {

// Thread-local message behind miq_last_error().
char* error_buffer();
int fail(int code, const char* fmt, ...);

// Developer A/B switches (tile shapes, kernel variants, timing-only builds: what tools/*_ab.py and the
// ablation tables under profiles/ are made with) are environment variables read per call through this one
// function.  Most choose between kernels that compute the same result; MIQ_GEMM_ABL selects timing-only
// builds of the GEMM kernel that leave work out and return WRONG results.  The product library is compiled
// with -DMIQ_NO_DEV_SWITCHES: this function then returns nullptr without reading the environment, and the
// timing-only kernels are not instantiated.  Only libmiq_dev.so (python -m quantization_amd.build --dev),
// which tools/ load, honours the switches.
inline const char* dev_switch(const char* name) {
#ifdef MIQ_NO_DEV_SWITCHES
    (void)name;
    return nullptr;
#else
    return getenv(name);
#endif
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

// What an entry point does with the sizes it is handed BEFORE any arithmetic on them (signed overflow is undefined
// behaviour, and a size query given nonsense must answer 0, not trap): every size, count, stride, padding in [0, 2^31),
// so that a product of two fits int64, and longer products through size_product().  The host code is built with
// -fsanitize=undefined for a test that hands every entry point such sizes (tests/test_host_sanitizer.py).
inline bool all_below_2_31(std::initializer_list<int64_t> v) {
    for (int64_t d : v)
        if (d < 0 || d >= (1ll << 31)) return false;
    return true;
}
// The product of non-negative sizes, or -1 when a factor is negative or the product reaches 2^62.
inline int64_t size_product(std::initializer_list<int64_t> f) {
    for (int64_t v : f)
        if (v < 0) return -1;
    int64_t p = 1;
    for (int64_t v : f) {
        if (v == 0) return 0;
        if (__builtin_mul_overflow(p, v, &p) || p >= (1ll << 62)) return -1;
    }
    return p;
}

// The same product, held at 2^60 where it would pass it: for geometry that is computed first and range-checked after
// (every byte count a kernel accepts is below 2^31, so a held value is refused like any other that is too large).
inline int64_t sat_product(std::initializer_list<int64_t> f) {
    const int64_t p = size_product(f);
    return p < 0 || p > (1ll << 60) ? (1ll << 60) : p;
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int64_t round_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

// hipGetLastError() after a launch -> MIQ_ELAUNCH with the runtime's text.
inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MIQ_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return MIQ_OK;
}

// Dynamic LDS above 64 KiB has to be opted into per kernel (and per device); cheap to repeat.
inline int allow_lds(const void* kernel, int bytes) {
    if (bytes <= 64 * 1024) return MIQ_OK;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return fail(MIQ_ELAUNCH, "hipFuncSetAttribute(LDS=%d): %s", bytes, hipGetErrorString(e));
    return MIQ_OK;
}

#define MIQ_REQUIRE(cond, code, ...)                  \
    do {                                              \
        if (!(cond)) return ::miq::fail(code, __VA_ARGS__); \
    } while (0)

}  // namespace miq
