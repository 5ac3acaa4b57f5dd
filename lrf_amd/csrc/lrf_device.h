// lrf_device.h — device-side helpers shared by the kernel files of liblrf_hip.so (vector typedefs, the stamps of diagnostic
// builds, the colour transform, the exact 64-leaf reduction tree, DPP broadcast operands).  Everything here is inline.
#ifndef LRF_DEVICE_H
#define LRF_DEVICE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>

#include "lrf_internal.h"

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned __attribute__((aligned(1))) u32_unaligned;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double d16 __attribute__((ext_vector_type(16)));

#define LRF_EPS 1e-16f
// Householder columns whose squared norm is at or below this are skipped (oracle: tridiagonalize): cascaded rounding noise of
// rank-deficient Gram matrices lands in the denormal range, where t = 2 / |v|^2 overflows
#define LRF_SIGMA_TINY 1e-280

// Diagnostic build only (-DLRF_STAMPS, never shipped): per-phase cycle sums of k_bcd, wave 0 of each workgroup.
#if defined(LRF_STAMPS) || defined(LRF_INIT_STAMPS) || defined(LRF_BLK_STAMPS) || defined(LRF_REG_STAMPS)
__device__ unsigned long long g_stamps[8 * 16384];
__device__ __forceinline__ unsigned long long stamp_now()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#endif
#ifdef LRF_STAMPS
#define STAMP(var) unsigned long long var = stamp_now()
#define STAMP_ADD(acc, a, b) acc += (b) - (a)
__device__ unsigned long long g_gsp[4 * 16384];
#ifndef LRF_GS_PROBE
#define LRF_GS_PROBE 0
#endif
static constexpr int getenv_probe_dummy = LRF_GS_PROBE;
#define GSP_ADD(slot, a, b)                                                                     \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 16384) atomicAdd(&g_gsp[4 * blockIdx.x + (slot)], (b) - (a))
#else
#define STAMP(var)
#define STAMP_ADD(acc, a, b)
#define GSP_ADD(slot, a, b)
#endif

__device__ __forceinline__ int reflect_idx(int i, int n)
{
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

// offset + einsum("ij,j...->i...") for one pixel: k-ordered fma chain from 0 (sgemm, K = 3), then offset + acc.
// ycc from three already-loaded channel bytes
__device__ __forceinline__ float ycc_of(float r, float g, float b, int c)
{
    const float T[3][3] = {{0.299f, 0.587f, 0.114f}, {-0.168736f, -0.331264f, 0.5f}, {0.5f, -0.418688f, -0.081312f}};
    float acc = 0.f;
    acc = fmaf(T[c][0], r, acc);
    acc = fmaf(T[c][1], g, acc);
    acc = fmaf(T[c][2], b, acc);
    return (c ? 128.f : 0.f) + acc;
}

// tree64 of the oracle: lane i ends with s[i] + s[i+off] for off = 32..1; lane 0 holds the result,
// which is broadcast.  (Lanes >= off compute unused values.)
// The partner fetches of the tree without the LDS crossbar: lane i needs lane i + off.
//   off = 32: v_permlane32_swap (upper half of one register <-> lower half of the other)
//   off = 16: v_permlane16_swap (odd 16-lane rows <-> even rows)
//   off <= 8: DPP row_shl inside the 16-lane row
// Only lanes < off need a correct partner, which is exactly what these give; lane 0 ends with the tree sum.
__device__ __forceinline__ double partner_32(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(b[1], a[1]); // second result: lanes 0-31 hold the former lanes 32-63
}
__device__ __forceinline__ double partner_16(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double(b[1], a[1]); // second result: even rows hold the former odd rows
}
template <int OFF>
__device__ __forceinline__ double partner_row(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x100 + OFF, 0xf, 0xf, true); // row_shl:OFF -> lane i reads lane i + OFF
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x100 + OFF, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_tree64(double v)
{
    v = v + partner_32(v);
    v = v + partner_16(v);
    v = v + partner_row<8>(v);
    v = v + partner_row<4>(v);
    v = v + partner_row<2>(v);
    v = v + partner_row<1>(v);
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// acc += b * u with a wave-uniform b (SGPR operand).  Spelled in assembly so that the SLP vectoriser does not pair the
// independent accumulators of the exact Gauss-Seidel into v_pk_fma_f32 (which costs thousands of register moves there).
__device__ __forceinline__ void fmac_su(float& acc, float b_uniform, float u)
{
    asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "s"(b_uniform), "v"(u));
}
// acc += tab[lane N of each 16-lane row] * x, and the broadcast alone: a wave-uniform table kept in VGPRs reaches the VALU
// through DPP row_newbcast with no memory latency at all (the same device k_bcd_w uses for V and its b table)
template <int N>
__device__ __forceinline__ void fmac_bc16(float& acc, float tab, float x)
{
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(tab), "v"(x), "n"(N));
}
template <int N>
__device__ __forceinline__ float get_bc16(float tab)
{
    float out;
    asm("v_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(out) : "v"(tab), "n"(N));
    return out;
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop the compiler cannot leave rolled (`#pragma unroll` is a
// request: inside k_bcd_p<true, 12> the operand loop of w32_block stayed a loop, its register arrays went to scratch memory)
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    [&]<int... Is>(std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }(std::make_integer_sequence<int, N>{});
}

// ---- memory policies of the BCD block bodies (w_block / w16_block / w32_block) -------------------------------------------------
// How a block reaches the data that OTHER waves produce or consume between two U updates: the V table, the b table, the
// int8 U rows and the partial tables of X^T u / u^T u.  (X is read-only: always plain loads.)
//   MemLaunch  launch-per-iteration kernels: a kernel boundary separates producer and consumer — plain loads and stores;
//   MemSc1     inside ONE persistent launch (k_bcd_p): every such byte is stored `sc1` (write-through: leaves the XCD's L2)
//              and loaded `sc1` (L1 bypass), the form MI355X_MICROARCH.md's inter-workgroup visibility section prices
//              (first row of its sc1 table; the storing wave drains with `s_waitcnt vmcnt(0)` before its ticket / flag).
struct MemLaunch {
    static constexpr bool kSc1 = false;
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ unsigned ld_u32(const unsigned* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
    static __device__ __forceinline__ void st_u32(void* p, unsigned v) { *reinterpret_cast<u32_unaligned*>(p) = v; } // any alignment
    static __device__ __forceinline__ void st_u8(int8_t* p, int8_t v) { *p = v; }
};
struct MemSc1 {
    static constexpr bool kSc1 = true;
    static __device__ __forceinline__ float ld(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    static __device__ __forceinline__ unsigned ld_u32(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    static __device__ __forceinline__ void st(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    static __device__ __forceinline__ void st_u32(void* p, unsigned v) // any alignment (an atomic store wants four)
    {
        asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    }
    static __device__ __forceinline__ void st_u8(int8_t* p, int8_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
};
#endif
