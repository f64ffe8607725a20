// Kernel launches of the inference path go through ESCX_LAUNCH: normally a plain hipLaunchKernelGGL; while the per-launch profiler is on
// (escx_profile_enable, PROF in escx_internal.h) the FIRST launch of a profiled group is made with hipExtLaunchKernelGGL, whose start / stop events carry
// the begin and end timestamps of the DISPATCH ITSELF - the same two timestamps rocprofv3's kernel trace reports - instead of a pair of marker events
// around the launch, which also time the gap between a marker and the dispatch (round 5: +35 ... 53 % under two-stream execution).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <tuple>
#include <utility>

namespace escx {
struct LaunchTimer { hipEvent_t start = nullptr, stop = nullptr; int launches = 0; };
extern thread_local LaunchTimer* g_launch_timer;     // escx_api.cpp; non-null only inside a PROF scope of a handle with profiling on
}  // namespace escx

namespace escx {
// hipExtLaunchKernelGGL with the implicit argument conversions of kernel<<<...>>>(...) (hip_ext.h's template insists on exact tuple construction)
template <typename... Formals, size_t... I>
inline void ext_launch_tuple(void (*kernel)(Formals...), dim3 grid, dim3 block, unsigned lds, hipStream_t s, hipEvent_t a, hipEvent_t b, std::tuple<Formals...>& t, std::index_sequence<I...>) {
    void* ptrs[sizeof...(Formals) ? sizeof...(Formals) : 1] = {const_cast<void*>(static_cast<const void*>(&std::get<I>(t)))...};
    (void)hipExtLaunchKernel(reinterpret_cast<const void*>(kernel), grid, block, ptrs, lds, s, a, b, 0);
}
template <typename... Formals, typename... Actuals>
inline void ext_launch(void (*kernel)(Formals...), dim3 grid, dim3 block, unsigned lds, hipStream_t s, hipEvent_t a, hipEvent_t b, Actuals&&... args) {
    static_assert(sizeof...(Formals) == sizeof...(Actuals), "kernel argument count");
    std::tuple<Formals...> t{static_cast<Formals>(std::forward<Actuals>(args))...};
    ext_launch_tuple(kernel, grid, block, lds, s, a, b, t, std::index_sequence_for<Formals...>{});
}
}  // namespace escx

#define ESCX_LAUNCH(kern, grid, block, lds, stream, ...) do { \
        ::escx::LaunchTimer* _lt = ::escx::g_launch_timer; \
        if (_lt && _lt->launches++ == 0) ::escx::ext_launch(kern, grid, block, lds, stream, _lt->start, _lt->stop, __VA_ARGS__); \
        else hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__); \
    } while (0)
