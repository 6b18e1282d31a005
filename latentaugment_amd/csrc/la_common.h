// Shared definitions for the latentaug HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LA_OK 0
#define LA_ERR_ARG (-1)        // bad argument / unsupported configuration
#define LA_ERR_HIP (-2)        // a HIP runtime call failed (see la_last_error)
#define LA_ERR_WORKSPACE (-3)  // caller-provided workspace too small

#define LA_WAVE 64

void la_set_error(const char* msg);

#define LA_CHECK_ARG(cond, msg)            \
    do {                                   \
        if (!(cond)) {                     \
            la_set_error(msg);             \
            return LA_ERR_ARG;             \
        }                                  \
    } while (0)

#define LA_CHECK_LAUNCH()                              \
    do {                                               \
        hipError_t e__ = hipGetLastError();            \
        if (e__ != hipSuccess) {                       \
            la_set_error(hipGetErrorString(e__));      \
            return LA_ERR_HIP;                         \
        }                                              \
    } while (0)

#define LA_HIP(call)                                   \
    do {                                               \
        hipError_t e__ = (call);                       \
        if (e__ != hipSuccess) {                       \
            la_set_error(hipGetErrorString(e__));      \
            return LA_ERR_HIP;                         \
        }                                              \
    } while (0)

// kernel classes of the launch profiler (la_prof.hip); the first four are the contraction classes
enum { LA_PC_CONV_HALO = 0, LA_PC_CONV_FLAT, LA_PC_CONV_SPLITK, LA_PC_CONV_F32, LA_PC_PRESPLIT, LA_PC_FIR, LA_PC_SEAM, LA_PC_TORGB,
       LA_PC_BANK, LA_PC_NCLASS };
bool la_prof_enabled();      // profiler active: callers keep their launches eager
// Development switches -- kernel-variant selectors for in-process A/B measurements (la_dev_knob_set) and LA_* environment switches --
// exist in the DEVELOPMENT build only (make dev: -DLA_DEV, liblatentaug_hip_dev.so, used by scripts/).  The product library has
// neither: every knob reads 0, no LA_* variable is looked at, la_dev_knob_set is not exported.
#ifdef LA_DEV
int la_dev_knob(int id);
const char* la_dev_env(const char* name);      // getenv
#else
static inline int la_dev_knob(int) { return 0; }
static inline const char* la_dev_env(const char*) { return nullptr; }
#endif
#define LA_KNOB_HALO_MF 0     // halo contraction form: 0 = default, 8 = round-2 form, else the MF bits of la_conv_bf16_halo_kernel
#define LA_KNOB_HALO_MING 1   // dev: grids of fewer points than this go to split-K instead of the halo kernel (0 = 1157: up to 34x34)
#define LA_KNOB_FLAT_MF 2     // flat / split-K contraction form: 0 = default (16x16x32 MFMA), 8 = 32x32x16
#define LA_KNOB_KSPLIT 3       // dev: force this many K slices on the split-K launches (0 = the cost model's choice)
#define LA_KNOB_HALO_STAMP 5   // dev: 1 = the MF 5 halo kernel records per-wave segment clocks (la_dev_dbg_read)
#define LA_KNOB_HALO_LDSPAD 6  // dev: extra KB of LDS per workgroup of the MF 5 halo kernel (occupancy experiments)
#define LA_NKNOB 16
int la_prof_open(int cls, double flops, double bytes, hipStream_t stream);     // -> slot or -1
void la_prof_close(int slot, hipStream_t stream);

static inline int la_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------- device helpers
__device__ __forceinline__ float la_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide sum for blockDim.x == 256 (4 waves).  `red` is >= 4 floats of LDS.  All threads get the result.
__device__ __forceinline__ float la_block_sum_256(float v, float* red) {
    v = la_wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// power-of-two operand scale of the fp16 split (la_conv_bf16.hip): brings a tensor's max magnitude into [2^14, 2^15)
__device__ __forceinline__ float la_pow2_scale(float amax) {
    if (!(amax > 0.f) || !isfinite(amax)) return 1.f;
    int e;
    frexpf(amax, &e);                    // amax = f * 2^e, f in [0.5, 1)
    int s = 15 - e;                      // scaled max in [2^14, 2^15): fp16 never overflows, 3 more bits above the subnormals
    s = s > 100 ? 100 : (s < -100 ? -100 : s);
    return ldexpf(1.f, s);
}

// Running per-sample fp16 operand scale handed from a producing kernel to the contraction that consumes its output: the slot
// starts a pass at LA_XS_INIT (2^100, the largest scale la_pow2_scale returns); every producing workgroup lowers it to
// la_pow2_scale(mult * its own max) -- positive floats order like unsigned ints, so an atomicMin on the bit pattern does it, the
// result is the scale of the overall maximum whatever the arrival order, and the consumer just reads a float.  `seen` is an
// earlier (possibly stale, i.e. larger) read of the slot: workgroups that cannot lower it skip the atomic.
// A slot is a ROW of LA_XS_SUBS sub-slots per sample, each in a 128-byte line of its own (LA_XS_FAN floats per row): a producing
// workgroup lowers the sub-slot picked by its id, the consumer takes the minimum of the row (la_xs_get) -- so that a launch of thousands
// of short workgroups finishing together (the split-K finish pass, the seam kernel) spreads its atomics over 32 LINES per sample
// instead of serialising on one (device-scope atomics execute at the memory side, one line at a time: round 2 measured the
// single-address form at 21 -> 65 us and 158 -> 281 us for those two and kept a reduction launch for them; round 4 measured 32
// sub-slots inside ONE line no better, +13 us / +40 us).  The minimum of the row is the scale of the overall maximum either way.
#define LA_XS_INIT 0x71800000u
#define LA_XS_SUBS 32
#define LA_XS_LINE 32                          // floats per 128-byte line
#define LA_XS_FAN (LA_XS_SUBS * LA_XS_LINE)    // floats per slot row
__device__ __forceinline__ float la_xs_peek(const float* slot) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(slot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void la_xs_lower(float* slot, float seen, float mult, float wg_max) {
    if (!(wg_max > 0.f)) return;
    const float s = la_pow2_scale(mult * wg_max);
    if (s < seen) atomicMin(reinterpret_cast<unsigned*>(slot), __float_as_uint(s));
}
// sub-slot of the calling workgroup inside a row (any spread will do; the multipliers keep neighbouring workgroups of every grid apart)
// (returned as the float offset of the sub-slot inside the row)
__device__ __forceinline__ int la_xs_sub(int salt = 0) { return (int)((blockIdx.x + 7u * blockIdx.y + 13u * blockIdx.z + (unsigned)salt) & (LA_XS_SUBS - 1)) * LA_XS_LINE; }
// final per-sample scale: plain arrays [B] (fan <= 1: scales computed by a reduction launch or known a priori) or slot rows [B][LA_XS_FAN]
__device__ __forceinline__ float la_xs_get(const float* p, int b, int fan) {
    if (fan <= 1) return p[b];
    const float* q = p + (long)b * LA_XS_FAN;
    float v[LA_XS_SUBS];
#pragma unroll
    for (int i = 0; i < LA_XS_SUBS; ++i) v[i] = q[i * LA_XS_LINE];      // (all in flight: one round trip)
    float m = v[0];
#pragma unroll
    for (int i = 1; i < LA_XS_SUBS; ++i) m = fminf(m, v[i]);
    return m;
}

// activation ids follow the reference's cuda_idx (bias_act.py:20-30): 1 linear, 2 relu, 3 lrelu
#define LA_ACT_LINEAR 1
#define LA_ACT_RELU 2
#define LA_ACT_LRELU 3

// y = clamp(act(v) * gain)  (bias_act.cu:23-147 semantics, grad=0)
__device__ __forceinline__ float la_act_fwd(float v, int act, float alpha, float gain, float clamp) {
    if (act == LA_ACT_LRELU) v = v > 0.f ? v : v * alpha;
    else if (act == LA_ACT_RELU) v = v > 0.f ? v : 0.f;
    v *= gain;
    if (clamp >= 0.f) v = fminf(fmaxf(v, -clamp), clamp);
    return v;
}

// d(y)/d(v) expressed through the saved OUTPUT y, as the reference's kernel does (bias_act.cu:46-48,71-72,141):
// slope from the sign of y, zero where the clamp is active.
__device__ __forceinline__ float la_act_bwd_from_y(float y, int act, float alpha, float gain, float clamp) {
    float s = gain;
    if (act == LA_ACT_LRELU) s = y > 0.f ? gain : gain * alpha;
    else if (act == LA_ACT_RELU) s = y > 0.f ? gain : 0.f;
    if (clamp >= 0.f && fabsf(y) >= clamp) s = 0.f;
    return s;
}

// pre-activation value (x + b) recovered from the saved output (only meaningful where the clamp is inactive)
__device__ __forceinline__ float la_act_inv(float y, int act, float alpha, float gain) {
    float v = y / gain;
    if (act == LA_ACT_LRELU && v < 0.f) v = v / alpha;
    return v;
}
